// la_gemm_core.h -- LDS-tiled MFMA main loop shared by every GEMM-shaped kernel.
//
//   D^T tile:  acc[mi][ni] (16x16 f32)  +=  W[n0+i][k] * A[m0+j][k]
//
// Both operands are K-contiguous ("NT" GEMM: nn.Linear weight layout [N][K] and
// row-major activations [M][K]), so A and W tiles are staged and read the same
// way.  One stage is 128 rows x 128 BYTES of K per operand (64 bf16 / 32 f32),
// written by global_load_lds_dwordx4 (16 B per lane, 8 rows per wave
// instruction).  The LDS image is linear; the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied to the per-lane global
// SOURCE column and again on the read (cdna guide rule 21):
//      slot' = slot ^ ((row >> 1) & 7)          (slot = 16-byte column, 0..7)
// A lane (r = lane & 15, q = lane >> 4) reads 16 B of row r at K-slot 4*ks + q:
//   bf16: 8 consecutive k  -> one v_mfma_f32_16x16x32_bf16 operand
//   f32 : 4 consecutive k  -> element i feeds the i-th of four v_mfma_f32_16x16x4_f32
//         (the k order inside a 16-wide group is permuted identically for A and W,
//          so every product a[m][k]*w[n][k] is formed exactly once).
// The MFMA is issued with W as the A-operand and the activations as the
// B-operand, so a lane ends up holding 4 CONSECUTIVE output columns n of one
// row m (8/16-byte epilogue stores, float4 residual loads).
#pragma once
#include "la_common.h"

namespace la {
namespace gemm {

constexpr int BN = 128;       // rows of W (output cols n) per workgroup
constexpr int BKB = 128;      // bytes of K per stage per row

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <typename T> struct KElems;
template <> struct KElems<bf16_t> { static constexpr int v = BKB / 2; };
template <> struct KElems<float> { static constexpr int v = BKB / 4; };
template <> struct KElems<_Float16> { static constexpr int v = BKB / 2; };

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// Stage ROWS rows x 128 B of one operand: ROWS/8 wave-instructions of 1 KiB, spread over NWAVES waves.
// `src` points at element [row0][k0] of a row-major matrix with `ld_bytes` pitch;
// rows are clamped to `last_row` (reads stay in bounds; the results of clamped rows are never stored).
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const unsigned char *src, int64_t ld_bytes, int row0, int last_row,
                                           unsigned char *lds_tile, int wave, int lane) {
    constexpr int PER_WAVE = ROWS / 8 / NWAVES;
    static_assert(PER_WAVE >= 1 && PER_WAVE * 8 * NWAVES == ROWS, "tile rows must split evenly over the waves");
    const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int piece = wave * PER_WAVE + i;
        const int rt = piece * 8 + r8;  // row inside the tile
        int row = row0 + rt;
        row = row > last_row ? last_row : row;
        const unsigned char *g = src + (int64_t)row * ld_bytes + ((slot ^ swz(rt)) << 4);
        unsigned char *l = lds_tile + piece * 1024;  // wave-uniform; hardware adds lane*16
        la::glds16(g, l);
    }
}

__device__ __forceinline__ uint4 read_frag(const unsigned char *lds_tile, int row, int slot) {
    return *reinterpret_cast<const uint4 *>(lds_tile + row * BKB + ((slot ^ swz(row)) << 4));
}

// ---- transposed operand (float32 only): the matrix is stored [K][rows] (rows contiguous), e.g. the activations X[m][k_in] as
// the "W operand" of dW[n][k_in] = sum_m dY[m][n] X[m][k_in], where the contraction index m is the slow one.  One stage =
// 32 k-rows x 128 rows x 4 B = the same 16 KiB; a wave instruction moves two k-rows of 512 B.  LDS image: k-row kk at
// kk*512, its 16-byte chunk c (4 consecutive rows) in slot c ^ g(kk), g(kk) = ((kk >> 2) & 3) << 2: a lane (r, q) then reads
// its four k = 16 ks + 4q + j as four ds_read_b32 at rows 16 i + r, and the four q groups land on disjoint bank quarters.
__device__ __forceinline__ int swz_t(int kk) { return ((kk >> 2) & 3) << 2; }

template <int NWAVES>
__device__ __forceinline__ void stage_tile_t(const unsigned char *src, int64_t ld_bytes, int k0, int last_k, int row0, int rows_total,
                                             unsigned char *lds_tile, int wave, int lane) {
    constexpr int PER_WAVE = 16 / NWAVES;          // 16 instructions of two k-rows each
    static_assert(PER_WAVE >= 1 && PER_WAVE * NWAVES == 16, "16 two-row pieces must split evenly over the waves");
    const int h = lane >> 5, s = lane & 31;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int piece = wave * PER_WAVE + i;
        const int kk = piece * 2 + h;
        int krow = k0 + kk;
        krow = krow > last_k ? last_k : krow;      // rows past K are re-reads of the last one; their products are masked
        int m = row0 + ((s ^ swz_t(kk)) << 2);
        m = m > rows_total - 4 ? rows_total - 4 : m;   // rows_total % 4 == 0 (host check): stays inside the k-row
        const unsigned char *g = src + (int64_t)krow * ld_bytes + (int64_t)m * 4;
        la::glds16(g, lds_tile + piece * 1024);
    }
}

// the four k = 16 ks + 4 q + j (j = 0..3) of tile row `row` -> same element order as read_frag's 16-byte K-slot 4 ks + q
__device__ __forceinline__ uint4 read_frag_t(const unsigned char *lds_tile, int row, int ks, int q) {
    const unsigned char *b = lds_tile + (ks * 16 + 4 * q) * 512 + ((((row >> 2) ^ (q << 2)) << 4) | ((row & 3) << 2));
    uint4 v;
    v.x = *reinterpret_cast<const unsigned *>(b);
    v.y = *reinterpret_cast<const unsigned *>(b + 512);
    v.z = *reinterpret_cast<const unsigned *>(b + 1024);
    v.w = *reinterpret_cast<const unsigned *>(b + 1536);
    return v;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
    }
};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <> struct Mma<_Float16> {
    __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
    }
};

// Tile configuration: WM waves along M (64 rows each) x 2 waves along N (64 columns each), STAGES-deep LDS ring.
//   <2,2>: 128x128 tile, 4 waves, 64 KiB  -> 2 workgroups per CU (small problems, f32 parity mode)
//   <4,3>: 256x128 tile, 8 waves, 144 KiB -> 1 workgroup per CU, two K-stages in flight across the barrier
template <int WM_, int STAGES_> struct Cfg {
    static constexpr int WM = WM_, STAGES = STAGES_;
    static constexpr int NW = WM * 2;
    static constexpr int THREADS = NW * 64;
    static constexpr int TM = WM * 64, TN = 128;
    static constexpr int STAGE = (TM + TN) * BKB;
    static constexpr int LDS = STAGES * STAGE;
    static constexpr int LOADS = (TM + TN) / 8 / NW;  // global_load_lds instructions per wave per stage
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N < 0, "add the literal");
}

// acc[mi][ni]: rows m = m0 + wm*64 + mi*16 + (lane & 15); cols n = n0 + wn*64 + ni*16 + (lane >> 4)*4 + reg
// TA / TW (float32 only): that operand is stored transposed, [K][rows] with pitch lda / ldw (see stage_tile_t).  With a
// transposed operand K need not be a multiple of the 32-element stage: the last stage's k >= K are masked out of the A
// fragments (a K-contiguous operand must then have a row pitch that covers the rounded-up K: its tail reads stay in the row).
template <typename T, typename C, bool TA = false, bool TW = false>
__device__ __forceinline__ void mainloop(const T *A, int64_t lda, int M, const T *W, int64_t ldw, int N, int K,
                                         int m0, int n0, unsigned char *lds, f32x4 (&acc)[4][4]) {
    static_assert((!TA && !TW) || sizeof(T) == 4, "transposed operands are built for float32");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    constexpr int KE = KElems<T>::v;
    const int nk = (K + KE - 1) / KE;
    const unsigned char *Ab = reinterpret_cast<const unsigned char *>(A);
    const unsigned char *Wb = reinterpret_cast<const unsigned char *>(W);
    const int64_t lda_b = lda * (int64_t)sizeof(T), ldw_b = ldw * (int64_t)sizeof(T);

#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kt, int buf) {
        unsigned char *st = lds + buf * C::STAGE;
        const int64_t koff = (int64_t)kt * BKB;
        if constexpr (TA) {
            static_assert(!TA || C::TM == 128, "transposed A: 128-row tiles");
            stage_tile_t<C::NW>(Ab, lda_b, kt * KE, K - 1, m0, M, st, wave, lane);
        } else {
            stage_tile<C::TM, C::NW>(Ab + koff, lda_b, m0, M - 1, st, wave, lane);
        }
        if constexpr (TW) stage_tile_t<C::NW>(Wb, ldw_b, kt * KE, K - 1, n0, N, st + C::TM * BKB, wave, lane);
        else stage_tile<C::TN, C::NW>(Wb + koff, ldw_b, n0, N - 1, st + C::TM * BKB, wave, lane);
    };
    auto compute = [&](int buf, int kvalid) {
        const unsigned char *st = lds + buf * C::STAGE;
        const unsigned char *at = st + (TA ? 0 : (wm * 64) * BKB);
        const unsigned char *wt = st + C::TM * BKB + (TW ? 0 : (wn * 64) * BKB);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (TA) af[i] = read_frag_t(at, wm * 64 + i * 16 + r, ks, q);
                else af[i] = read_frag(at, i * 16 + r, ks * 4 + q);
                if constexpr (TW) wf[i] = read_frag_t(wt, wn * 64 + i * 16 + r, ks, q);
                else wf[i] = read_frag(wt, i * 16 + r, ks * 4 + q);
            }
            if constexpr (TA || TW) {
                if (kvalid < KE) {                 // last, partial stage: k = 16 ks + 4 q + j >= kvalid contributes nothing
                    const int kb = ks * 16 + 4 * q;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (kb + 0 >= kvalid) af[i].x = 0u;
                        if (kb + 1 >= kvalid) af[i].y = 0u;
                        if (kb + 2 >= kvalid) af[i].z = 0u;
                        if (kb + 3 >= kvalid) af[i].w = 0u;
                    }
                }
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<T>::run(wf[ni], af[mi], acc[mi][ni]);
        }
    };

    if constexpr (C::STAGES == 2) {
        // two barriers' worth of drain per K-step; the second workgroup on the CU covers the bubbles
        stage(0, 0);
        wait_vmcnt<0>();
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
            compute(cur, K - kt * KE);
            wait_vmcnt<0>();
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // 3-deep ring, ONE raw barrier per K-step, the loads of the next two K-steps stay in flight across it
        // (counted vmcnt: cdna guide "Pipelining across barriers").  Iteration kt:
        //   wait until stage kt has landed (<= LOADS younger loads outstanding) -> barrier (also: every wave is
        //   done reading stage kt-1) -> refill that buffer with stage kt+2 -> fragments + MFMAs of stage kt.
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        int cur = 0, nxt2 = 2;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) wait_vmcnt<C::LOADS>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage(kt + 2, nxt2);
            compute(cur, K - kt * KE);
            cur = cur == 2 ? 0 : cur + 1;
            nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
        }
    }
}

// XCD-aware tile order: consecutive workgroup ids land on different XCDs
// (round-robin dispatch), so give each XCD a contiguous run of tiles that
// share W panels / A panels in its private L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
}

// L2-aware tile order on top of xcd_remap: N is cut into groups of `group` column tiles whose W panels
// (group * 128 rows * K) fit an XCD's 4 MiB L2 next to the streaming A panels; inside a group the order is
// n fastest, then m.  Without it every row of tiles re-reads ALL of W from beyond L2 (measured: 26 % L2 misses,
// 2.1 GB fetched for a 0.4 GB problem).
struct TileCoord { int tm, tn; };
__device__ __forceinline__ TileCoord tile_coord(int tile, int tiles_m, int tiles_n, int group) {
    const int per_group = group * tiles_m;
    const int g = tile / per_group;
    const int base_n = g * group;
    const int gw = min(group, tiles_n - base_n);
    const int rem = tile - g * per_group;
    return TileCoord{rem / gw, base_n + rem % gw};
}
// The same with the M dimension cut into blocks of `mblock` row tiles first (0 = the order above): inside a block the column
// groups follow each other, so a block's A panels (mblock x 256 rows x K) are re-read once per group while they are still in the
// XCD's L2 / the Infinity Cache instead of once per sweep over ALL of M; with mblock x group = the 32 workgroups an XCD runs at a
// time, the resident set is an mblock x group supertile walked in step along K.
__device__ __forceinline__ TileCoord tile_coord_mb(int tile, int tiles_m, int tiles_n, int group, int mblock) {
    if (mblock <= 0) return tile_coord(tile, tiles_m, tiles_n, group);
    const int per_block = mblock * tiles_n;
    int b = tile / per_block;
    const int nb = tiles_m / mblock;
    int rows = mblock;
    if (b >= nb) { b = nb; rows = tiles_m - nb * mblock; }
    const int off = tile - b * per_block;
    const int g = off / (rows * group);
    const int base_n = g * group;
    const int gw = min(group, tiles_n - base_n);
    const int rem = off - g * rows * group;
    return TileCoord{b * mblock + rem / gw, base_n + rem % gw};
}
inline int pick_group(int K, int elem_bytes, int tiles_n) {
    const char *force = la::dev_env("LA_GEMM_GROUP");   // experiment build only: developer sweep, read per launch
    if (force) { const int g = atoi(force); return g < 1 ? 1 : (g > tiles_n ? tiles_n : g); }
    const int64_t tile_bytes = (int64_t)BN * K * elem_bytes;
    int g = (int)((2 << 20) / tile_bytes);
    g = g < 1 ? 1 : (g > 16 ? 16 : g);
    return g > tiles_n ? tiles_n : g;
}

}  // namespace gemm
}  // namespace la
