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

constexpr int BM = 128;       // rows of A (output rows m) per workgroup
constexpr int BN = 128;       // rows of W (output cols n) per workgroup
constexpr int BKB = 128;      // bytes of K per stage per row
constexpr int NTHREADS = 256; // 4 waves: 2 (m) x 2 (n), 64 x 64 outputs each
constexpr int STAGE_BYTES = (BM + BN) * BKB;  // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // double buffered

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <typename T> struct KElems;
template <> struct KElems<bf16_t> { static constexpr int v = BKB / 2; };
template <> struct KElems<float> { static constexpr int v = BKB / 4; };

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// Stage one 128-row operand tile: 16 wave-instructions of 1 KiB, 4 per wave.
// `src` points at element [row0][k0] of a row-major matrix with `ld_bytes` pitch;
// rows are clamped to `last_row` (reads stay in bounds; the results of clamped rows are never stored).
__device__ __forceinline__ void stage_tile(const unsigned char *src, int64_t ld_bytes, int row0, int last_row,
                                           unsigned char *lds_tile, int wave, int lane) {
    const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (wave * 4 + i) * 8 + r8;  // row inside the tile
        int row = row0 + rt;
        row = row > last_row ? last_row : row;
        const unsigned char *g = src + (int64_t)row * ld_bytes + ((slot ^ swz(rt)) << 4);
        unsigned char *l = lds_tile + (wave * 4 + i) * 1024;  // wave-uniform; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)l, 16, 0, 0);
    }
}

__device__ __forceinline__ uint4 read_frag(const unsigned char *lds_tile, int row, int slot) {
    return *reinterpret_cast<const uint4 *>(lds_tile + row * BKB + ((slot ^ swz(row)) << 4));
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
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

// acc[mi][ni]: rows m = m0 + wm*64 + mi*16 + (lane & 15); cols n = n0 + wn*64 + ni*16 + (lane >> 4)*4 + reg
template <typename T>
__device__ __forceinline__ void mainloop(const T *A, int64_t lda, int M, const T *W, int64_t ldw, int N, int K,
                                         int m0, int n0, unsigned char *lds, f32x4 (&acc)[4][4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    constexpr int KE = KElems<T>::v;
    const int nk = K / KE;
    const unsigned char *Ab = reinterpret_cast<const unsigned char *>(A);
    const unsigned char *Wb = reinterpret_cast<const unsigned char *>(W);
    const int64_t lda_b = lda * (int64_t)sizeof(T), ldw_b = ldw * (int64_t)sizeof(T);

#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_tile(Ab, lda_b, m0, M - 1, lds, wave, lane);
    stage_tile(Wb, ldw_b, n0, N - 1, lds + BM * BKB, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        unsigned char *st = lds + cur * STAGE_BYTES;
        if (kt + 1 < nk) {
            unsigned char *nx = lds + (cur ^ 1) * STAGE_BYTES;
            const int64_t koff = (int64_t)(kt + 1) * BKB;
            stage_tile(Ab + koff, lda_b, m0, M - 1, nx, wave, lane);
            stage_tile(Wb + koff, ldw_b, n0, N - 1, nx + BM * BKB, wave, lane);
        }
        const unsigned char *at = st + (wm * 64) * BKB;
        const unsigned char *wt = st + BM * BKB + (wn * 64) * BKB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = read_frag(at, i * 16 + r, ks * 4 + q);
                wf[i] = read_frag(wt, i * 16 + r, ks * 4 + q);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<T>::run(wf[ni], af[mi], acc[mi][ni]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
}

// XCD-aware tile order: consecutive workgroup ids land on different XCDs
// (round-robin dispatch), so give each XCD a contiguous run of tiles that
// share W panels / A panels in its private L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
}

}  // namespace gemm
}  // namespace la
