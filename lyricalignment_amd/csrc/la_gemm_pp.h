// la_gemm_pp.h -- 256x256 "ping-pong" bf16 GEMM main loop for gfx950 (large-M encoder GEMMs).
//
// Why: the 128x128 kernel moves 64 flop per byte of L2->LDS traffic and measured ~10 TB/s of that traffic at
// 650 TFLOP/s -- more than half of what global_load_lds delivers from L2 chip-wide (MI355X_MICROARCH "Indexed rows").
// A 256x256 tile halves the traffic per flop (128 flop/B).  One 512-thread workgroup per CU (128 KiB LDS):
//   8 waves = 2 (M) x 4 (N); wave (wr, wc) owns rows wr*128..+127 and columns wc*64..+63 (8 x 4 MFMA 16x16 tiles,
//   128 accumulator VGPRs).  K advances in tiles of 64 (128 B rows), two LDS buffers of 64 KiB (A 32 KiB | W 32 KiB).
//
// Schedule (cdna guide, 8-phase template and "Two waves per SIMD"): each K-tile is four phases, one accumulator
// quadrant (64 rows x 32 cols, 16 MFMAs) per phase; a phase is  [LOAD segment: ds_read fragments + 2 global_load_lds]
// s_barrier [COMPUTE segment: 16 MFMAs] s_barrier.  The two wave groups wr=0 / wr=1 (which share SIMDs pairwise) run
// staggered by one barrier, so one group's LOAD segment sits beside the other group's COMPUTE segment: the matrix pipe
// sees back-to-back MFMA clusters while LDS reads and DMA issue hide under the partner's MFMAs.
//   quadrant order per K-tile: (a0,b0) (a0,b1) (a1,b1) (a1,b0)  -> fragment reads: 8+4, 4, 8, 0 ds_read_b128
// Prefetch: the loads of K-tile t+2 go into the buffer K-tile t occupied, W part first (dead after phase 1 of tile t),
// then the A part (dead after phase 2), 2 instructions per phase; they are retired by ONE counted s_waitcnt vmcnt per
// K-tile, placed before the barrier that precedes the first read of that buffer (hazard analysis in DESIGN.md).
#pragma once
#include <type_traits>

#include "la_gemm_core.h"

namespace la {
namespace gemm {

struct PP {
    static constexpr int TM = 256, TN = 256, THREADS = 512;
    static constexpr int OPB = 256 * BKB;       // one operand of one K-tile: 32 KiB
    static constexpr int BUF = 2 * OPB;         // 64 KiB
    static constexpr int LDS = 2 * BUF;         // 128 KiB
};

// piece p (0..31) of an operand tile = rows 8p..8p+7; each wave stages 4 pieces of W and 4 of A per K-tile
__device__ __forceinline__ void pp_stage_piece(const unsigned char *src, int64_t ld_bytes, int row0, int last_row,
                                               unsigned char *lds_op, int piece, int lane) {
    const int rt = piece * 8 + (lane >> 3);
    int row = row0 + rt;
    row = row > last_row ? last_row : row;
    const unsigned char *g = src + (int64_t)row * ld_bytes + (((lane & 7) ^ swz(rt)) << 4);
    la::glds16(g, lds_op + piece * 1024);
}

// raw barrier fenced for the COMPILER on both sides (memory ops and, via sched_barrier, MFMAs stay in their segment)
#define LA_PP_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_s_barrier();        \
        asm volatile("" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// acc[mi][ni]: rows m = m0 + wr*128 + mi*16 + (lane & 15); cols n = n0 + wc*64 + ni*16 + (lane >> 4)*4 + reg
template <typename T16 = bf16_t>
__device__ __forceinline__ void mainloop_pp(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                            int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches, no exec masking
    const int wr = wave >> 2, wc = wave & 3;       // group = wr: waves 0-3 / 4-7 pair up on the SIMDs
    const int r = lane & 15, q = lane >> 4;
    const int nk = K / 64;
    const unsigned char *Ab = reinterpret_cast<const unsigned char *>(A);
    const unsigned char *Wb = reinterpret_cast<const unsigned char *>(W);
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;

#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's 4 pieces of each operand tile: pieces 4*wave .. 4*wave+3, two per issue slot.  Per-lane byte offsets
    // (row clamp + source-side swizzle) are loop invariant; the K advance lives in the scalar base pointer.
    unsigned voff_a[4], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * wave + i) * 8 + (lane >> 3);
        const int sw = ((lane & 7) ^ swz(rt)) << 4;
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_a[i] = (unsigned)((int64_t)ra * lda_b - (int64_t)m0 * lda_b) + sw;   // relative to row m0: stays < 2^32
        voff_w[i] = (unsigned)((int64_t)rw * ldw_b - (int64_t)n0 * ldw_b) + sw;
    }
    const unsigned char *a_row0 = Ab + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = Wb + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    // the pair (half) of this wave's four pieces of the W (A) tile of K-tile kt
    auto issue_w = [&](int kt, int half) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            glds16_so(voff_w[2 * half + j], w_row0 + (int64_t)kt * BKB, lds0 + (kt & 1) * PP::BUF + PP::OPB + (4 * wave + 2 * half + j) * 1024);
    };
    auto issue_a = [&](int kt, int half) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            glds16_so(voff_a[2 * half + j], a_row0 + (int64_t)kt * BKB, lds0 + (kt & 1) * PP::BUF + (4 * wave + 2 * half + j) * 1024);
    };

    // ---- prologue: K-tiles 0 and 1 completely (the steady-state schedule starts with tile 2) ----
    issue_w(0, 0); issue_w(0, 1); issue_a(0, 0); issue_a(0, 1);
    if (nk > 1) { issue_w(1, 0); issue_w(1, 1); issue_a(1, 0); issue_a(1, 1); }
    // only K-tile 0 has to be here for the first MFMA; tile 1's eight pieces stay in flight and are retired by the counted
    // waits that close iteration 0 (they are older than everything those waits leave outstanding)
    if (nk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    LA_PP_BARRIER();
    if (wr == 1) LA_PP_BARRIER();   // stagger: group 1 runs one barrier behind group 0

    uint4 af[8], b0f[4], b1f[4];    // a: 4 row tiles x 2 k-steps; b0 / b1: 2 column tiles x 2 k-steps each
    auto read_a = [&](const unsigned char *abuf, int half) {   // rows wr*128 + half*64 + mi*16 + r
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[mi * 2 + ks] = read_frag(abuf, wr * 128 + half * 64 + mi * 16 + r, ks * 4 + q);
    };
    auto read_b = [&](const unsigned char *wbuf, int half, uint4 (&bf)[4]) {   // W rows wc*64 + half*32 + ni*16 + r
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf[ni * 2 + ks] = read_frag(wbuf, wc * 64 + half * 32 + ni * 16 + r, ks * 4 + q);
    };
    auto mma_quadrant = [&](int ah, int bh, const uint4 (&bf)[4]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    Mma<T16>::run(bf[ni * 2 + ks], af[mi * 2 + ks], acc[ah * 4 + mi][bh * 2 + ni]);
    };

    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char *abuf = lds + (kt & 1) * PP::BUF;
        const unsigned char *wbuf = abuf + PP::OPB;
        const bool pf2 = kt + 2 < nk;
        const bool pf1 = kt >= 1 && kt + 1 < nk;
        // Issue slots while tile kt is computed (2 instructions per phase and wave, at the end of the LOAD segment; a buffer
        // half is refilled only after the barrier that follows the completion of its last fragment read by BOTH groups):
        //   group 0:  ph0 W half 1 of kt+1 | ph1 A half 0 of kt+1 | ph2 A half 1 of kt+1 | ph3 W half 0 of kt+2
        //   group 1:  ph0 A half 0 of kt+1 | ph1 A half 1 of kt+1 | ph2 W half 0 of kt+2 | ph3 W half 1 of kt+2
        auto dma = [&](int ph) {
            if (wr == 0) {
                if (ph == 0) { if (pf1) issue_w(kt + 1, 1); }
                else if (ph == 1) { if (pf1) issue_a(kt + 1, 0); }
                else if (ph == 2) { if (pf1) issue_a(kt + 1, 1); }
                else { if (pf2) issue_w(kt + 2, 0); }
            } else {
                if (ph == 0) { if (pf1) issue_a(kt + 1, 0); }
                else if (ph == 1) { if (pf1) issue_a(kt + 1, 1); }
                else if (ph == 2) { if (pf2) issue_w(kt + 2, 0); }
                else { if (pf2) issue_w(kt + 2, 1); }
            }
        };
        // ---------------- phase 0: quadrant (a0, b0) ----------------
        read_a(abuf, 0);
        read_b(wbuf, 0, b0f);
        dma(0);
        LA_PP_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quadrant(0, 0, b0f);
        LA_PP_BARRIER();
        // ---------------- phase 1: quadrant (a0, b1) ----------------
        read_b(wbuf, 1, b1f);
        dma(1);
        LA_PP_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quadrant(0, 1, b1f);
        LA_PP_BARRIER();
        // ---------------- phase 2: quadrant (a1, b1) ----------------
        read_a(abuf, 1);
        dma(2);
        LA_PP_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quadrant(1, 1, b1f);
        LA_PP_BARRIER();
        // ---------------- phase 3: quadrant (a1, b0) ----------------
        dma(3);
        if (wr == 1) {
            // g1's LOAD segment closes with the barrier that precedes g0's first read of tile kt+1: retire everything
            // of tile kt+1 (issued >= 2 segments ago); only tile kt+2's W pieces issued so far may stay in flight
            // (the 2 of phase 2 + the 2 of this segment)
            if (!pf2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        LA_PP_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        mma_quadrant(1, 0, b0f);
        if (wr == 0) {
            // g0's COMPUTE segment closes with the same barrier: only tile kt+2's 2 W instructions may stay in flight
            if (pf2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        LA_PP_BARRIER();
    }
    if (wr == 0) LA_PP_BARRIER();   // re-align the groups
}

// ---------------------------------------------------------------------------------------------------------------------
// 64-byte-row stages (one k-step of 32 16-bit elements per row) used by the hand-placed loops below: a DMA piece is 16 rows x
// 64 B (lane l -> row l >> 2, 16-byte chunk l & 3); chunk c of row r sits in slot c ^ ((-(r >> 2)) & 3), which makes the 16-row
// fragment reads conflict-free for ds_read_b128's lane groups (checked exhaustively against the bank rule of
// MI355X_MICROARCH.md: 4 LDS cycles per read).
__device__ __forceinline__ int swz2(int row) { return (-(row >> 2)) & 3; }

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF> __device__ __forceinline__ void ds_read128_asm(u32x4 &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// "duo", hand-placed (LA_PP_DBG=81): the ping-pong kernel's geometry -- 8 waves = 2 (M) x 4 (N), 128x64 wave tiles, two waves
// per SIMD -- with the k2 stages (K = 32, ring of 4) and the hand-placed instruction stream of mainloop_mono_asm instead of
// the LOAD / COMPUTE role alternation: every wave runs the same program, per k-step its 32 MFMAs with the next k-step's 12
// fragment reads (4 of W, 8 of A; j = 1, 3, .., 23), its 4 LDS-DMA pieces of stage s + 4 (j = 2, 10, 18, 26), one counted wait
// (j = 28) and one barrier (j = 29) in their gaps.  The partner wave of the SIMD covers a wave's DMA-issue and barrier time
// with its own MFMAs.  Accumulators in VGPRs (128) + two fragment sets (96): the epilogue is the ping-pong kernel's.
//   Timeline as in mainloop_mono_asm (prefetch distance 4 on the ring of 4; vmcnt counts are per wave: 4 pieces per stage).
template <typename T> struct MmaAsmV;
template <> struct MmaAsmV<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &w, const u32x4 &a, f32x4 &acc) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
    }
};
template <> struct MmaAsmV<_Float16> {
    __device__ static __forceinline__ void run(const u32x4 &w, const u32x4 &a, f32x4 &acc) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
    }
};

// Where the reads / DMA pieces / wait / barrier sit among the 32 MFMAs was A/B-tested in round 2 (reads one per MFMA from j = 0,
// DMA as an early burst, wait / barrier at the very end, the two waves of a SIMD in different gaps): all within +-1 % on the
// encoder's shapes, one 2.5 % slower -- the loop is LDS-bandwidth-bound (96 KiB of fragment reads + 32 KiB of DMA writes per
// k-step = the 128 B per clock of a 1024-cycle k-step), not placement-bound.  This is the placement that was kept.
// Diagnostic build (LA_EXTRA_CXXFLAGS=-DLA_TILE_STAMPS, tools/tile_timeline.py): the 100 MHz wall clock at the end of the prologue
#ifdef LA_TILE_STAMPS
#define LA_STAMP_PARAM , unsigned long long &stamp_t1
#define LA_STAMP_T1() stamp_t1 = __builtin_amdgcn_s_memrealtime()
#else
#define LA_STAMP_PARAM
#define LA_STAMP_T1()
#endif
// STAT_WC >= 0 (the LayerNorm-consumer GEMMs, round 4): the wave also takes the row statistics of the A rows it multiplies -- the
// RAW rows of the residual stream whose LayerNorm is folded into this GEMM -- from the fragments it already holds: per k-step
// v_dot2c_f32_{bf16,f16} of each fragment dword with (1, 1) and with itself, i.e. 16 two-term dot products into sacc = (sum, sum of
// squares) of row blocks mi = 2 STAT_WC and 2 STAT_WC + 1 (the four waves of a row group hold the same A fragments: wave column
// wc = STAT_WC takes a quarter of them, so every row is summed once per tile; STAT_WC is a template parameter because the
// fragment registers are named at compile time -- the kernel switches on its wave column around the whole main loop).  Two per slot
// in eight MFMA gaps that carry nothing else.  Replaces the separate statistics pass over the stream (la_row_stats16: 98 MB and a
// launch per LayerNorm) and the epilogue's statistics loads.
template <typename T> struct Dot2cAsm;
template <> struct Dot2cAsm<bf16_t> {
    static constexpr unsigned ONES = 0x3F803F80u;
    __device__ static __forceinline__ void run(float &s1, float &s2, unsigned v, unsigned ones) {
        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1) : "v"(v), "v"(ones));
        asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(s2) : "v"(v));
    }
};
template <> struct Dot2cAsm<_Float16> {
    static constexpr unsigned ONES = 0x3C003C00u;
    __device__ static __forceinline__ void run(float &s1, float &s2, unsigned v, unsigned ones) {
        asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(s1) : "v"(v), "v"(ones));
        asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(s2) : "v"(v));
    }
};
// The loop in three pieces, so that a PERSISTENT workgroup (gemm_pp_persist_kernel) can issue the next tile's first stages before the
// current tile's epilogue: duo_setup (per-tile addresses), duo_issue_prologue (the DMA of stages 0 .. 3), duo_run (entry wait, first
// fragments, the k-steps).  mainloop_duo_asm = the three in a row.
struct DuoCtx {
    unsigned voff[4];              // this wave's four DMA pieces of a stage: per-lane byte offsets (row clamp + source-side swizzle)
    const unsigned char *src0;     // the wave's operand panel (W rows n0 .. for waves 0-3, A rows m0 .. for waves 4-7), K offset 0
    unsigned piece0;               // LDS address of the wave's first piece in ring slot 0
    unsigned fa_lo, fa_hi, fw_lo, fw_hi;   // fragment read bases (slots 0-1 / 2-3)
    // SEGMENTED K (the f16x2 products, la_f32x2.hip): an operand row holds two PLANES of Kc elements (hi terms, lo terms of a float32
    // value split into two halves; plane pitch below), and the k loop walks 3 x seg_stages stages -- per 32-wide chunk of K the
    // products (lo, hi), (hi, lo), (hi, hi) -- so stage t of the flat loop reads plane seg_off[t % 3] of this wave's operand at column (t / 3) * 32
    // (duo_stage_src).
    int seg_stages = 0;
    unsigned seg_off0 = 0, seg_off1 = 0;   // byte offset of the first / second segment's plane inside a row (this wave's operand; the third reads plane 0)
};
// source of stage t (32 elements of K = 64 bytes per row) of this wave's operand panel: wave-uniform arithmetic
template <bool SEG>
__device__ __forceinline__ const unsigned char *duo_stage_src(const DuoCtx &c, int t) {
    if constexpr (!SEG) {
        return c.src0 + (int64_t)t * 64;
    } else {
        // (masks, not selects between the context's fields: hipcc turns a select of two loads from the context into an indexed load
        //  and leaves the whole context in scratch)
        const int n = c.seg_stages;
        // Round 6: the three products of one 32-wide k chunk run back to back -- stage t = chunk t / 3, product t % 3 = (lo, hi), (hi, lo),
        // (hi, hi) -- so that the second read of the chunk's a_hi / w_hi tiles follows the first within two stages (an L2 hit) instead of a whole
        // pass over K later: QKV 768 -> 752, MLP-up 1025 -> 995, out-proj 276 -> 270 us at 48000 rows, K = 4096 flat (profiles/r6_x2_interleave.txt;
        // until round 5 each product ran over all of K in turn, small terms first; the error against float64 stays below the float32 kernel's own).
        // -DLA_X2_SEGMENT_MAJOR keeps the old order (A/B partner).
        (void)n;
#ifdef LA_X2_SEGMENT_MAJOR
        const int seg = (t >= n ? 1 : 0) + (t >= 2 * n ? 1 : 0);
        const int col = t - seg * n;
#else
        const int col = t / 3;
        const int seg = t - 3 * col;
#endif
        const unsigned off = ((unsigned)-(int)(seg == 0) & c.seg_off0) | ((unsigned)-(int)(seg == 1) & c.seg_off1);
        // (wave-uniform by construction; the DMA's base operand must sit in SGPRs, so say so to the compiler)
        const uint64_t u = (uint64_t)(uintptr_t)(c.src0 + off + (int64_t)col * 64);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return reinterpret_cast<const unsigned char *>((uintptr_t)(((uint64_t)hi << 32) | lo));
    }
}
template <typename T16>
__device__ __forceinline__ void duo_setup(DuoCtx &c, const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int m0, int n0,
                                          unsigned lds0, int wave, int lane) {
    constexpr int STAGE = 32768, OPS = 16384, SB = 64;
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
    // this wave's 4 pieces of every stage: waves 0..3 carry the W image (rows 64 wave .. + 63), waves 4..7 the A image
    const bool is_w = wave < 4;
    const int pw = wave & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * pw + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int rg = (is_w ? n0 : m0) + rt;
        const int lim = (is_w ? N : M) - 1;
        rg = rg > lim ? lim : rg;
        c.voff[i] = (unsigned)((int64_t)(rg - (is_w ? n0 : m0)) * (is_w ? ldw_b : lda_b)) + sw;
    }
    c.src0 = is_w ? reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b
                  : reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    c.piece0 = lds0 + (is_w ? OPS : 0) + (4 * pw) * 1024;
    c.fa_lo = lds0 + (unsigned)((wr * 128 + r) * SB + ((q ^ swz2(r)) << 4));
    c.fa_hi = c.fa_lo + 2 * STAGE;
    c.fw_lo = lds0 + OPS + (unsigned)((wc * 64 + r) * SB + ((q ^ swz2(r)) << 4));
    c.fw_hi = c.fw_lo + 2 * STAGE;
}
// Prefetch distance of the hand-placed loop in stages (ring of 4): 4 = stage s + 4 goes into the slot k-step s has just freed;
// 3 (a library built with -DLA_DUO_DIST=3: one stage less in flight per wave) is the A/B partner, same results.
#ifndef LA_DUO_DIST
#define LA_DUO_DIST 4
#endif
template <bool SEG = false>
__device__ __forceinline__ void duo_issue_prologue(const DuoCtx &c) {
    constexpr int STAGE = 32768;
#pragma unroll
    for (int st = 0; st < LA_DUO_DIST; ++st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_so(c.voff[i], duo_stage_src<SEG>(c, st), c.piece0 + st * STAGE + i * 1024);
    }
}
// PREFETCHED: stages 0 .. 3 were issued BEFORE the previous tile's epilogue (whose >= 24 vector-memory operations -- its stores --
// are younger than them in this wave's queue): one wait retires all four without forcing a single store, k-steps 0 and 1 (which
// would retire stages 2 and 3) wait for their fragment reads only, and from k-step 2 on -- whose wait needs stage 4, issued
// after those stores -- the stream is the ordinary one: by then (>= 2 k-steps, ~2 us) the stores have drained.
// Timing-only knock-outs of the hand-placed k-step (a library built with LA_EXTRA_CXXFLAGS=-DLA_DUO_PROBE=<bits>; results are garbage):
// 1 = no barrier, 2 = no counted vmcnt wait, 4 = no LDS-DMA issue, 8 = no fragment reads.  What each leg costs the loop.
#ifndef LA_DUO_PROBE
#define LA_DUO_PROBE 0
#endif
template <typename T16 = bf16_t, int STAT_WC = -1, bool PREFETCHED = false, bool SEG = false>
__device__ __forceinline__ void duo_run(const DuoCtx &c, int K, f32x4 (&acc)[8][4] LA_STAMP_PARAM, float *sacc = nullptr) {
    constexpr int STAGE = 32768;
    const int ns = K / 32;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned fa_lo = c.fa_lo, fa_hi = c.fa_hi, fw_lo = c.fw_lo, fw_hi = c.fw_hi;
    const unsigned piece0 = c.piece0;
    unsigned voff[4] = {c.voff[0], c.voff[1], c.voff[2], c.voff[3]};
    auto issue1 = [&](const unsigned char *src, int slot, int i) __attribute__((always_inline)) {
        glds16_so(voff[i], src, piece0 + slot * STAGE + i * 1024);
    };
    // (PREFETCHED -- the persistent kernel -- has its entry waits counted for distance 4: not to be run from a -DLA_DUO_DIST=3 build)
    if constexpr (PREFETCHED) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (LA_DUO_DIST == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // stages 0 and 1 landed; the 8 pieces of stages 2 and 3 may stay in flight
    LA_PP_BARRIER();
    LA_STAMP_T1();
    u32x4 fa[2][8], fw[2][4];
    ds_read128_asm<0 * 1024>(fw[0][0], fw_lo); ds_read128_asm<1 * 1024>(fw[0][1], fw_lo); ds_read128_asm<2 * 1024>(fw[0][2], fw_lo);
    ds_read128_asm<3 * 1024>(fw[0][3], fw_lo);
    ds_read128_asm<0 * 1024>(fa[0][0], fa_lo); ds_read128_asm<1 * 1024>(fa[0][1], fa_lo); ds_read128_asm<2 * 1024>(fa[0][2], fa_lo);
    ds_read128_asm<3 * 1024>(fa[0][3], fa_lo); ds_read128_asm<4 * 1024>(fa[0][4], fa_lo); ds_read128_asm<5 * 1024>(fa[0][5], fa_lo);
    ds_read128_asm<6 * 1024>(fa[0][6], fa_lo); ds_read128_asm<7 * 1024>(fa[0][7], fa_lo);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[1][i] = fa[0][i];
#pragma unroll
    for (int i = 0; i < 4; ++i) fw[1][i] = fw[0][i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();

    unsigned stat_ones = Dot2cAsm<T16>::ONES;
    if constexpr (STAT_WC >= 0) asm volatile("" : "+v"(stat_ones));          // one VGPR for the loop, not a literal per instruction
    auto kstep = [&](int s, auto nxc, auto pfc, auto vmc, auto curc, auto slotc) __attribute__((always_inline)) {
        constexpr bool NX = decltype(nxc)::value, PF = decltype(pfc)::value;
        constexpr int VM = decltype(vmc)::value;
        constexpr int CUR = decltype(curc)::value, SLOT = decltype(slotc)::value;
        constexpr int SN = (SLOT + 1) & 3, SW = (SLOT + LA_DUO_DIST) & 3;
        constexpr int OFFN = (SN & 1) * STAGE;
        const unsigned fan = SN >= 2 ? fa_hi : fa_lo, fwn = SN >= 2 ? fw_hi : fw_lo;
        const unsigned char *src = duo_stage_src<SEG>(c, s + LA_DUO_DIST);
        static_for<0, 32>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, mi = j >> 2, ni = j & 3;
            MmaAsmV<T16>::run(fw[CUR][ni], fa[CUR][mi], acc[mi][ni]);
            if constexpr (STAT_WC >= 0) {
                constexpr int slot = j == 4 ? 0 : j == 6 ? 1 : j == 12 ? 2 : j == 14 ? 3 : j == 20 ? 4 : j == 22 ? 5 : j == 25 ? 6 : j == 27 ? 7 : -1;
                if constexpr (slot >= 0) {
                    constexpr int fi = slot >> 2, dw = slot & 3;         // slots 0-3: the dwords of fragment 0, slots 4-7: of fragment 1
                    Dot2cAsm<T16>::run(sacc[2 * fi], sacc[2 * fi + 1], fa[CUR][2 * STAT_WC + fi][dw], stat_ones);
                }
            }
            if constexpr (NX && (j & 1) == 1 && j / 2 < 12 && !(LA_DUO_PROBE & 8)) {
                constexpr int i = j / 2;                        // fragments of the next k-step: W 0..3, then A 0..7
                if constexpr (i < 4) ds_read128_asm<OFFN + i * 1024>(fw[CUR ^ 1][i], fwn);
                else ds_read128_asm<OFFN + (i - 4) * 1024>(fa[CUR ^ 1][i - 4], fan);
            }
            if constexpr (PF && (j & 7) == 2 && !(LA_DUO_PROBE & 4)) issue1(src, SW, j >> 3);
            if constexpr (j == 28) {
                if constexpr (VM == 63 || (LA_DUO_PROBE & 2)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (PREFETCHED: its stage landed long ago)
                else if constexpr (VM == 8 && LA_DUO_DIST == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else if constexpr (VM == 8 || VM == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            if constexpr (j == 29 && !(LA_DUO_PROBE & 1)) asm volatile("s_barrier" ::: "memory");
        });
    };
    typedef std::false_type F;
    typedef std::true_type TT;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    typedef std::integral_constant<int, 4> V4;
    typedef std::integral_constant<int, 8> V8;
    typedef std::integral_constant<int, 63> V63;
    int s = 0;                              // ns is a multiple of 4, >= 8 (host check)
    if constexpr (PREFETCHED) {             // the first ring turn with the two wait-free k-steps (s + 4 < ns holds: ns >= 8)
        kstep(0, TT{}, TT{}, V63{}, I0{}, I0{});
        kstep(1, TT{}, TT{}, V63{}, I1{}, I1{});
        kstep(2, TT{}, TT{}, V8{}, I0{}, I2{});
        kstep(3, TT{}, TT{}, V8{}, I1{}, I3{});
        s = 4;
    }
    for (; s + 4 < ns; s += 4) {
        kstep(s, TT{}, TT{}, V8{}, I0{}, I0{});
        kstep(s + 1, TT{}, TT{}, V8{}, I1{}, I1{});
        kstep(s + 2, TT{}, TT{}, V8{}, I0{}, I2{});
        kstep(s + 3, TT{}, TT{}, V8{}, I1{}, I3{});
    }
    if constexpr (LA_DUO_DIST == 3) kstep(s, TT{}, TT{}, V4{}, I0{}, I0{});       // (stage ns - 1 is still to be issued)
    else kstep(s, TT{}, F{}, V4{}, I0{}, I0{});
    kstep(s + 1, TT{}, F{}, I0{}, I1{}, I1{});
    kstep(s + 2, TT{}, F{}, I0{}, I0{}, I2{});
    kstep(s + 3, F{}, F{}, I0{}, I1{}, I3{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
}
template <typename T16 = bf16_t, int STAT_WC = -1>
__device__ __forceinline__ void mainloop_duo_asm(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                                 int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4] LA_STAMP_PARAM,
                                                 float *sacc = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    DuoCtx c;
    duo_setup<T16>(c, A, lda, M, W, ldw, N, m0, n0, lds0, wave, lane);
    duo_issue_prologue(c);
#ifdef LA_TILE_STAMPS
    duo_run<T16, STAT_WC, false>(c, K, acc, stamp_t1, sacc);
#else
    duo_run<T16, STAT_WC, false>(c, K, acc, sacc);
#endif
}

// The same loop over SEGMENTED K (DuoCtx): A rows = [hi plane | lo plane] of Kc elements each at pitch plane_a (elements), W rows
// likewise at plane_w; computes sum_k (a_lo w_hi + a_hi w_lo + a_hi w_hi) in that order in one accumulation of 3 Kc / 32 stages.
template <typename T16>
__device__ __forceinline__ void mainloop_duo_seg_asm(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int Kc,
                                                     int64_t plane_a, int64_t plane_w, int m0, int n0, unsigned char *lds,
                                                     f32x4 (&acc)[8][4]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    DuoCtx c;
    duo_setup<T16>(c, A, lda, M, W, ldw, N, m0, n0, lds0, wave, lane);
    c.seg_stages = Kc / 32;
    c.seg_off0 = wave < 4 ? 0u : (unsigned)(plane_a * 2);          // W: hi, lo, hi
    c.seg_off1 = wave < 4 ? (unsigned)(plane_w * 2) : 0u;          // A: lo, hi, hi
    duo_issue_prologue<true>(c);
#ifdef LA_TILE_STAMPS
    unsigned long long stamp_unused = 0;
    duo_run<T16, -1, false, true>(c, 3 * Kc, acc, stamp_unused);
#else
    duo_run<T16, -1, false, true>(c, 3 * Kc, acc);
#endif
}

}  // namespace gemm
}  // namespace la
