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
// DBG (developer probes, LA_PP_DBG): bit0 = no in-loop DMA, bit1 = no MFMA, bit2 = s_setprio around the MFMA clusters
template <int DBG, typename T16 = bf16_t>
__device__ __forceinline__ void mainloop_pp(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                            int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4], unsigned *stamp_out = nullptr) {
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
    // one DMA piece: j = 0 / 1 of the pair (half) of this wave's four pieces of the W (A) tile of K-tile kt
    auto issue_w1 = [&](int kt, int half, int j) {
        if ((DBG & 1) && kt >= 2) return;
        const unsigned dst = lds0 + (kt & 1) * PP::BUF + PP::OPB + (4 * wave + 2 * half + j) * 1024;
        glds16_so(voff_w[2 * half + j], w_row0 + (int64_t)kt * BKB, dst);
    };
    auto issue_a1 = [&](int kt, int half, int j) {
        if ((DBG & 1) && kt >= 2) return;
        const unsigned dst = lds0 + (kt & 1) * PP::BUF + (4 * wave + 2 * half + j) * 1024;
        glds16_so(voff_a[2 * half + j], a_row0 + (int64_t)kt * BKB, dst);
    };
    auto issue_w = [&](int kt, int half) { issue_w1(kt, half, 0); issue_w1(kt, half, 1); };
    auto issue_a = [&](int kt, int half) { issue_a1(kt, half, 0); issue_a1(kt, half, 1); };
    // HEADN (developer A/B, LA_PP_DBG = 40 / 48): how many of a phase's two pieces are issued at the HEAD of the COMPUTE
    // segment -- after the barrier, under the s_waitcnt lgkmcnt(0) that waits for the fragment reads anyway -- instead of at
    // the end of the LOAD segment.  Later issue only relaxes the WAR conditions; the RAW waits keep their places and count
    // what is still to come (group 1's wait precedes its phase-3 COMPUTE head).
    constexpr int HEADN = (DBG == 40) ? 1 : (DBG == 48 ? 2 : 0);

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
        if (DBG & 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(af[i].x), "v"(af[i].y), "v"(af[i].z), "v"(af[i].w));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(bf[i].x), "v"(bf[i].y), "v"(bf[i].z), "v"(bf[i].w));
            return;
        }
        if (DBG & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    Mma<T16>::run(bf[ni * 2 + ks], af[mi * 2 + ks], acc[ah * 4 + mi][bh * 2 + ni]);
        if (DBG & 4) __builtin_amdgcn_s_setprio(0);
    };

#ifdef LA_PP_STAMPS
    // diagnostic build only (python -m lyricalignment_amd.build with LA_EXTRA_CXXFLAGS=-DLA_PP_STAMPS): s_memtime at the six
    // points of every phase of ONE K-tile (kt == 6), written to the buffer whose address LA_STAMP_PTR names
    unsigned st[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) st[i] = 0;
#define LA_STAMP(i) do { if (kt == 6) st[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define LA_STAMP(i) do { } while (0)
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char *abuf = lds + (kt & 1) * PP::BUF;
        const unsigned char *wbuf = abuf + PP::OPB;
        const bool pf2 = kt + 2 < nk;
        const bool pf1 = kt >= 1 && kt + 1 < nk;
        // Issue slots while tile kt is computed (2 instructions per phase and wave; a buffer half is refilled only
        // after the barrier that follows the completion of its last fragment read by BOTH groups):
        //   group 0:  ph0 W half 1 of kt+1 | ph1 A half 0 of kt+1 | ph2 A half 1 of kt+1 | ph3 W half 0 of kt+2
        //   group 1:  ph0 A half 0 of kt+1 | ph1 A half 1 of kt+1 | ph2 W half 0 of kt+2 | ph3 W half 1 of kt+2
        auto piece = [&](int ph, int j) {          // piece j (0 / 1) of this wave's phase-ph slot
            if (wr == 0) {
                if (ph == 0) { if (pf1) issue_w1(kt + 1, 1, j); }
                else if (ph == 1) { if (pf1) issue_a1(kt + 1, 0, j); }
                else if (ph == 2) { if (pf1) issue_a1(kt + 1, 1, j); }
                else { if (pf2) issue_w1(kt + 2, 0, j); }
            } else {
                if (ph == 0) { if (pf1) issue_a1(kt + 1, 0, j); }
                else if (ph == 1) { if (pf1) issue_a1(kt + 1, 1, j); }
                else if (ph == 2) { if (pf2) issue_w1(kt + 2, 0, j); }
                else { if (pf2) issue_w1(kt + 2, 1, j); }
            }
        };
        auto dma_load = [&](int ph) {              // the pieces of the slot that go at the end of the LOAD segment
#pragma unroll
            for (int j = 0; j < 2 - HEADN; ++j) piece(ph, j);
        };
        auto dma_head = [&](int ph) {              // ... and those at the head of the COMPUTE segment
#pragma unroll
            for (int j = 2 - HEADN; j < 2; ++j) piece(ph, j);
        };
        const bool dma_first = (DBG == 32) && (wc & 1);      // developer A/B: odd wave columns issue their DMA before their reads
        // ---------------- phase 0: quadrant (a0, b0) ----------------
        LA_STAMP(0);
        if (dma_first) dma_load(0);
        read_a(abuf, 0);
        read_b(wbuf, 0, b0f);
        LA_STAMP(1);
        if (!dma_first) dma_load(0);
        LA_STAMP(2);
        LA_PP_BARRIER();
        LA_STAMP(3);
        dma_head(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        LA_STAMP(4);
        mma_quadrant(0, 0, b0f);
        LA_STAMP(5);
        LA_PP_BARRIER();
        // ---------------- phase 1: quadrant (a0, b1) ----------------
        LA_STAMP(6);
        if (dma_first) dma_load(1);
        read_b(wbuf, 1, b1f);
        LA_STAMP(7);
        if (!dma_first) dma_load(1);
        LA_STAMP(8);
        LA_PP_BARRIER();
        LA_STAMP(9);
        dma_head(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        LA_STAMP(10);
        mma_quadrant(0, 1, b1f);
        LA_STAMP(11);
        LA_PP_BARRIER();
        // ---------------- phase 2: quadrant (a1, b1) ----------------
        LA_STAMP(12);
        if (dma_first) dma_load(2);
        read_a(abuf, 1);
        LA_STAMP(13);
        if (!dma_first) dma_load(2);
        LA_STAMP(14);
        LA_PP_BARRIER();
        LA_STAMP(15);
        dma_head(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        LA_STAMP(16);
        mma_quadrant(1, 1, b1f);
        LA_STAMP(17);
        LA_PP_BARRIER();
        // ---------------- phase 3: quadrant (a1, b0) ----------------
        LA_STAMP(18);
        LA_STAMP(19);
        dma_load(3);
        if (wr == 1) {
            // g1's LOAD segment closes with the barrier that precedes g0's first read of tile kt+1: retire everything
            // of tile kt+1 (issued >= 2 segments ago); only tile kt+2's W pieces issued so far may stay in flight
            // (the 2 of phase 2 + the 2 - HEADN of this segment)
            if (!pf2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (HEADN == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (HEADN == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        LA_STAMP(20);
        LA_PP_BARRIER();
        LA_STAMP(21);
        dma_head(3);
        __builtin_amdgcn_sched_barrier(0);
        LA_STAMP(22);
        mma_quadrant(1, 0, b0f);
        LA_STAMP(23);
        if (wr == 0) {
            // g0's COMPUTE segment closes with the same barrier: only tile kt+2's 2 W instructions may stay in flight
            if (pf2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        LA_PP_BARRIER();
        LA_STAMP(24);
    }
#ifdef LA_PP_STAMPS
    if (stamp_out && lane == 0 && blockIdx.x < 64) {
        unsigned *o = stamp_out + ((size_t)blockIdx.x * 8 + wave) * 32;
#pragma unroll
        for (int i = 0; i < 25; ++i) o[i] = st[i];
    }
#endif
    if (wr == 0) LA_PP_BARRIER();   // re-align the groups
}

// ---------------------------------------------------------------------------------------------------------------------
// "k2" main loop: the same 256x256 tile / 2 (M) x 4 (N) wave layout / ping-pong of the two wave groups, cut differently:
//   * a PHASE is one k-step (K = 32) of the wave's whole 128x64 tile: LOAD = 8 A + 4 W fragment reads (ds_read_b128) +
//     this wave's 4 DMA pieces, COMPUTE = 32 MFMAs (512 matrix cycles).  Every phase has the same 12 reads (the quadrant
//     schedule above has 12 / 4 / 8 / 0 per 16 MFMAs: its first LOAD segment alone is 48 KiB of LDS reads per wave
//     group = 192 LDS cycles + latency against a 256-cycle partner segment) and there are 4 barriers per K = 64, not 8.
//   * a STAGE is one k-step of both operands: 256 rows x 64 B each (32 KiB), ring of 4 stages (128 KiB).  A DMA piece is
//     16 rows x 64 B (lane l -> row l >> 2, 16-byte chunk l & 3); chunk c of row r sits in slot c ^ ((-(r >> 2)) & 3),
//     which makes the 16-row fragment reads conflict-free for ds_read_b128's lane groups (checked exhaustively against
//     the bank rule of MI355X_MICROARCH.md: 4 LDS cycles per read).
//   * prefetch distance 2: in LOAD(s) a wave issues its pieces of stage s + 2 into the slot stage s - 2 occupied.
//     Time in segments (one per barrier), group 0: LOAD(s) = 2s, COMPUTE(s) = 2s + 1; group 1 one segment later.
//     WAR: the last reads of stage s - 2 (group 1, LOAD(s - 2) = segment 2s - 3) were retired by its lgkmcnt(0) at the start
//       of segment 2s - 2; the refill is issued in segment 2s (group 0) / 2s + 1 (group 1): >= 2 barriers later.
//     RAW: stage s + 1 is first read in segment 2s + 2 (group 0).  Every wave retires its stage-(s + 1) pieces with one
//       counted vmcnt before the barrier that ends segment 2s + 1 -- group 0 at the end of COMPUTE(s), group 1 at the end
//       of LOAD(s) -- leaving only the 4 pieces of stage s + 2 in flight; the read follows one barrier after that wait.
template <int NST_> struct K2 {
    static constexpr int SB = 64;             // bytes of K per stage and row (32 x 16-bit)
    static constexpr int OPS = 256 * SB;      // one operand of one stage: 16 KiB
    static constexpr int STAGE = 2 * OPS;     // 32 KiB
    static constexpr int NST = NST_;          // ring slots: 4 (128 KiB, prefetch distance 2) or 5 (all 160 KiB of the CU, distance 3)
    static constexpr int LDS = NST * STAGE;
};
__device__ __forceinline__ int swz2(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ uint4 read_frag2(const unsigned char *op, int row, int q) {
    return *reinterpret_cast<const uint4 *>(op + row * 64 + ((q ^ swz2(row)) << 4));
}
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N == 0 || N == 4 || N == 8 || N == 12, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// Prefetch distance DIST = NST - 2 stages: in LOAD(s) a wave issues its pieces of stage s + DIST into the slot stage
// s + DIST - NST = s - 2 occupied (WAR above).  More stages in flight = more bytes in flight per CU: with a DMA latency of
// ~1 us under load, 64 KiB in flight cap the CU's L2 -> LDS rate near 50 GB/s, which is what the 2-buffer loop measures
// with the MFMAs removed; the wait that retires stage s + 1 then leaves 4 (DIST - 1) younger pieces in flight.
// VAR (developer A/B): bit0 = DMA issue before the fragment reads of a LOAD segment; bit1 = s_setprio 1 on waves 4-7
template <int VAR, int NST, typename T16 = bf16_t>
__device__ __forceinline__ void mainloop_k2(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                            int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4]) {
    typedef K2<NST> C;
    constexpr int DIST = NST - 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int ns = K / 32;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's pieces of every stage: rows 32 wave .. 32 wave + 31 of A and of W (two 16-row pieces each)
    unsigned voff_a[2], voff_w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = (2 * wave + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_a[i] = (unsigned)((int64_t)(ra - m0) * lda_b) + sw;      // relative to row m0: stays < 2^32
        voff_w[i] = (unsigned)((int64_t)(rw - n0) * ldw_b) + sw;
    }
    const unsigned char *a_row0 = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    auto issue = [&](int st, int slot) {     // this wave's 4 pieces of stage st into ring slot `slot`
        const unsigned dst = lds0 + slot * C::STAGE + (2 * wave) * 1024;
        const unsigned char *sa = a_row0 + (int64_t)st * C::SB, *sw = w_row0 + (int64_t)st * C::SB;
        glds16_so(voff_w[0], sw, dst + C::OPS);
        glds16_so(voff_a[0], sa, dst);
        glds16_so(voff_w[1], sw, dst + C::OPS + 1024);
        glds16_so(voff_a[1], sa, dst + 1024);
    };

#pragma unroll
    for (int st = 0; st < DIST; ++st)
        if (st < ns) issue(st, st);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
    if (wr == 1) LA_PP_BARRIER();   // stagger: group 1 runs one barrier behind group 0
    if ((VAR & 2) && wr == 1) __builtin_amdgcn_s_setprio(1);

    int slot_r = 0, slot_w = DIST;               // ring slots of stage s (read) and of stage s + DIST (refill)
    for (int s = 0; s < ns; ++s) {
        const unsigned char *abuf = lds + slot_r * C::STAGE + (wr * 128) * C::SB;
        const unsigned char *wbuf = lds + slot_r * C::STAGE + C::OPS + (wc * 64) * C::SB;
        const bool pf = s + DIST < ns;
        // pieces this wave may leave in flight when it retires stage s + 1: those of stages s + 2 .. min(s + DIST, ns - 1)
        const int ahead = (ns - 1 < s + DIST ? ns - 1 : s + DIST) - (s + 1);
        auto retire = [&]() {
            if (ahead >= 2 && DIST >= 3) wait_vm<8>();
            else if (ahead == 1) wait_vm<4>();
            else wait_vm<0>();
        };
        uint4 af[8], bf[4];
        // ---------------- LOAD(s) ----------------
        if ((VAR & 1) && pf) issue(s + DIST, slot_w);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = read_frag2(wbuf, ni * 16 + r, q);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) af[mi] = read_frag2(abuf, mi * 16 + r, q);
        if (!(VAR & 1) && pf) issue(s + DIST, slot_w);
        if (wr == 1) retire();   // group 1: its LOAD segment ends with the barrier that precedes group 0's first read of stage s + 1
        LA_PP_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- COMPUTE(s) ----------------
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) Mma<T16>::run(bf[ni], af[mi], acc[mi][ni]);
        if (wr == 0) retire();   // group 0: its COMPUTE segment ends with that same barrier
        LA_PP_BARRIER();
        slot_r = slot_r + 1 == NST ? 0 : slot_r + 1;
        slot_w = slot_w + 1 == NST ? 0 : slot_w + 1;
    }
    if ((VAR & 2) && wr == 1) __builtin_amdgcn_s_setprio(0);
    if (wr == 0) LA_PP_BARRIER();   // re-align the groups
}

// "k2f": the k2 stages WITHOUT the ping-pong -- all eight waves run the same program, ONE barrier per k-step, fragment reads
// and MFMAs interleaved by the compiler inside a wave and by the hardware between the two waves of a SIMD.  Per stage s:
//   counted vmcnt (own pieces of stage s landed; those of stages s + 1 .. s + DIST - 1 stay in flight)  ->  barrier (every
//   wave's pieces of stage s are in LDS; every wave's reads of stage s - 1 were consumed by MFMAs issued before it)  ->
//   refill the slot of stage s - 1 with stage s + DIST (so NST = DIST + 1 slots)  ->  12 fragment reads, 32 MFMAs.
template <int NST, typename T16 = bf16_t>
__device__ __forceinline__ void mainloop_k2f(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                             int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4]) {
    typedef K2<NST> C;
    constexpr int DIST = NST - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int ns = K / 32;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned voff_a[2], voff_w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = (2 * wave + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_a[i] = (unsigned)((int64_t)(ra - m0) * lda_b) + sw;
        voff_w[i] = (unsigned)((int64_t)(rw - n0) * ldw_b) + sw;
    }
    const unsigned char *a_row0 = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    auto issue = [&](int st, int slot) {
        const unsigned dst = lds0 + slot * C::STAGE + (2 * wave) * 1024;
        const unsigned char *sa = a_row0 + (int64_t)st * C::SB, *sw = w_row0 + (int64_t)st * C::SB;
        glds16_so(voff_w[0], sw, dst + C::OPS);
        glds16_so(voff_a[0], sa, dst);
        glds16_so(voff_w[1], sw, dst + C::OPS + 1024);
        glds16_so(voff_a[1], sa, dst + 1024);
    };
#pragma unroll
    for (int st = 0; st < DIST; ++st)
        if (st < ns) issue(st, st);
    int slot_r = 0, slot_w = DIST;
    for (int s = 0; s < ns; ++s) {
        // pieces younger than stage s that this wave has issued so far: stages s + 1 .. min(s + DIST - 1, ns - 1)
        const int ahead = (ns - 1 < s + DIST - 1 ? ns - 1 : s + DIST - 1) - s;
        if (ahead >= 3 && DIST >= 4) wait_vm<12>();
        else if (ahead >= 2 && DIST >= 3) wait_vm<8>();
        else if (ahead == 1) wait_vm<4>();
        else wait_vm<0>();
        LA_PP_BARRIER();
        if (s + DIST < ns) issue(s + DIST, slot_w);
        const unsigned char *abuf = lds + slot_r * C::STAGE + (wr * 128) * C::SB;
        const unsigned char *wbuf = lds + slot_r * C::STAGE + C::OPS + (wc * 64) * C::SB;
        uint4 af[8], bf[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = read_frag2(wbuf, ni * 16 + r, q);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) af[mi] = read_frag2(abuf, mi * 16 + r, q);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) Mma<T16>::run(bf[ni], af[mi], acc[mi][ni]);
        slot_r = slot_r + 1 == NST ? 0 : slot_r + 1;
        slot_w = slot_w + 1 == NST ? 0 : slot_w + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();      // every wave is done with the ring before the epilogue reuses the LDS
}

// "k2p" (LA_PP_DBG=80): the k2f stages with the fragment reads SOFTWARE-PIPELINED inside each wave -- a second fragment
// register set (48 VGPRs) takes the 12 reads of k-step s + 1 while the 32 MFMAs of k-step s issue, and the wave's 4 DMA pieces
// are spread over the same MFMAs (source order per group of 4 MFMAs: 1-2 reads, every other group one piece).  No LOAD
// segment: a wave's stream is MFMA-paced, the other wave of the SIMD covers its stalls, one barrier per k-step.
//   iteration s: vmcnt (own pieces of stage s + 1 landed; stage s + 2's four may fly) -> barrier (stage s + 1 visible to all;
//   every wave has finished iteration s - 1, so the slot of stage s -- its fragments were read during iteration s - 1 -- and
//   of every older stage is free) -> refill slot (s + 3) % 4 = slot of stage s - 1 with stage s + 3, read stage s + 1's fragments,
//   MFMAs of stage s.
template <typename T16 = bf16_t>
__device__ __forceinline__ void mainloop_k2p(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                             int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4]) {
    typedef K2<4> C;
    constexpr int NST = 4, DIST = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int ns = K / 32;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned voff_a[2], voff_w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = (2 * wave + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_a[i] = (unsigned)((int64_t)(ra - m0) * lda_b) + sw;
        voff_w[i] = (unsigned)((int64_t)(rw - n0) * ldw_b) + sw;
    }
    const unsigned char *a_row0 = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    auto issue1 = [&](int st, int slot, int i) {      // piece i (0, 1: W; 2, 3: A) of this wave's share of stage st
        const unsigned dst = lds0 + slot * C::STAGE + (i < 2 ? C::OPS : 0) + (2 * wave + (i & 1)) * 1024;
        if (i < 2) glds16_so(voff_w[i], w_row0 + (int64_t)st * C::SB, dst);
        else glds16_so(voff_a[i - 2], a_row0 + (int64_t)st * C::SB, dst);
    };
    auto frag_a = [&](int slot, int mi) { return read_frag2(lds + slot * C::STAGE + (wr * 128) * C::SB, mi * 16 + r, q); };
    auto frag_w = [&](int slot, int ni) { return read_frag2(lds + slot * C::STAGE + C::OPS + (wc * 64) * C::SB, ni * 16 + r, q); };
#pragma unroll
    for (int st = 0; st < DIST; ++st)
        if (st < ns) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue1(st, st, i);
        }
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
    uint4 fa[2][8], fw[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[0][i] = frag_a(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) fw[0][i] = frag_w(0, i);
    int slot_n = 1, slot_w = DIST % NST;
    // (a branch-free steady state -- the flags as template parameters -- lets hipcc software-pipeline harder: 256 VGPRs and
    // 422 spilled registers, half the speed; with the run-time flags it stays at 237 VGPRs, no spills)
    auto kstep = [&](int s, auto curc) {
        constexpr int cur = decltype(curc)::value;
        const int ahead = (ns - 1 < s + DIST - 1 ? ns - 1 : s + DIST - 1) - (s + 1);   // stages younger than s + 1 already issued
        if (s + 1 < ns) {
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        LA_PP_BARRIER();
        const bool pf = s + DIST < ns, nx = s + 1 < ns;
#pragma unroll
        for (int g = 0; g < 8; ++g) {       // 8 groups of 4 MFMAs: row tile g x the 4 column tiles
            if (nx) {
                fa[cur ^ 1][g] = frag_a(slot_n, g);
                if (g < 4) fw[cur ^ 1][g] = frag_w(slot_n, g);
            }
            if (pf && (g & 1) == 0) issue1(s + DIST, slot_w, g >> 1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) Mma<T16>::run(fw[cur][ni], fa[cur][g], acc[g][ni]);
        }
        slot_n = slot_n + 1 == NST ? 0 : slot_n + 1;
        slot_w = slot_w + 1 == NST ? 0 : slot_w + 1;
    };
    for (int s = 0; s < ns; s += 2) {
        kstep(s, std::integral_constant<int, 0>{});
        if (s + 1 < ns) kstep(s + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
}

// "half" main loop: a 128 x 256 tile for a 4-wave workgroup (waves 1 (M) x 4 (N), the same 128 x 64 wave tile), TWO such
// workgroups per CU (72 KiB of LDS and <= 256 VGPRs each).  The two workgroups of a CU are not synchronised with each
// other, so one's prologue, barrier waits and -- above all -- epilogue (7-30 k cycles per tile, un-overlapped when the CU
// holds a single 8-wave workgroup) run beside the other's MFMAs.  Price: the W stage is staged once per workgroup, so the
// L2 -> LDS traffic per flop is 1.5 x the 256 x 256 tile's (96 KiB per 2 x 64 K-steps per CU; tools/loadpath_bench.hip measures
// 116 GB/s per CU for this staging form against the ~45 GB/s the 256 x 256 loop draws).
// Stages: one k-step (K = 32): A 128 rows x 64 B (8 KiB) | W 256 rows x 64 B (16 KiB); ring of 3 (72 KiB), prefetch distance 2,
// one barrier per stage (the k2f scheme); per wave and stage 2 A pieces + 4 W pieces.
struct KH {
    static constexpr int TM = 128, TN = 256, THREADS = 256;
    static constexpr int SB = 64, OPA = 128 * SB, OPW = 256 * SB, STAGE = OPA + OPW, NST = 3, LDS = NST * STAGE;
};
template <int N> __device__ __forceinline__ void wait_vm6() {
    static_assert(N == 0 || N == 6, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}
template <typename T16 = bf16_t>
__device__ __forceinline__ void mainloop_half(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                              int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4]) {
    typedef KH C;
    constexpr int DIST = C::NST - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..3 = wc
    const int r = lane & 15, q = lane >> 4;
    const int ns = K / 32;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned voff_a[2], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * wave + i) * 16 + (lane >> 2);            // W rows 64 wave .. + 63
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_w[i] = (unsigned)((int64_t)(rw - n0) * ldw_b) + (((lane & 3) ^ swz2(rt)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = (2 * wave + i) * 16 + (lane >> 2);            // A rows 32 wave .. + 31
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        voff_a[i] = (unsigned)((int64_t)(ra - m0) * lda_b) + (((lane & 3) ^ swz2(rt)) << 4);
    }
    const unsigned char *a_row0 = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    auto issue = [&](int st, int slot) {
        const unsigned da = lds0 + slot * C::STAGE + (2 * wave) * 1024, dw = lds0 + slot * C::STAGE + C::OPA + (4 * wave) * 1024;
        const unsigned char *sa = a_row0 + (int64_t)st * C::SB, *sw = w_row0 + (int64_t)st * C::SB;
        glds16_so(voff_w[0], sw, dw);
        glds16_so(voff_a[0], sa, da);
        glds16_so(voff_w[1], sw, dw + 1024);
        glds16_so(voff_w[2], sw, dw + 2048);
        glds16_so(voff_a[1], sa, da + 1024);
        glds16_so(voff_w[3], sw, dw + 3072);
    };
#pragma unroll
    for (int st = 0; st < DIST; ++st)
        if (st < ns) issue(st, st);
    int slot_r = 0, slot_w = DIST;
    for (int s = 0; s < ns; ++s) {
        if (s + 1 < ns) wait_vm6<6>(); else wait_vm6<0>();      // own pieces of stage s landed; stage s + 1's six stay in flight
        LA_PP_BARRIER();
        if (s + DIST < ns) issue(s + DIST, slot_w);
        const unsigned char *abuf = lds + slot_r * C::STAGE;
        const unsigned char *wbuf = lds + slot_r * C::STAGE + C::OPA + (wave * 64) * C::SB;
        uint4 af[8], bf[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = read_frag2(wbuf, ni * 16 + r, q);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) af[mi] = read_frag2(abuf, mi * 16 + r, q);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) Mma<T16>::run(bf[ni], af[mi], acc[mi][ni]);
        slot_r = slot_r + 1 == C::NST ? 0 : slot_r + 1;
        slot_w = slot_w + 1 == C::NST ? 0 : slot_w + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
}

// Same 256x256 tile / wave layout with the plain structure: ONE barrier per K-tile, the next tile's 8 DMA pieces per
// wave issued right after it, fragment reads and MFMAs left to the compiler's interleave (2 waves per SIMD cover each
// other's LDS latency).  Kept as the A/B partner of the ping-pong schedule (LA_PP_DBG=8).
__device__ __forceinline__ void mainloop_flat256(const bf16_t *A, int64_t lda, int M, const bf16_t *W, int64_t ldw, int N, int K,
                                                 int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int nk = K / 64;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned voff_a[4], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * wave + i) * 8 + (lane >> 3);
        const int sw = ((lane & 7) ^ swz(rt)) << 4;
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_a[i] = (unsigned)((int64_t)(ra - m0) * lda_b) + sw;
        voff_w[i] = (unsigned)((int64_t)(rw - n0) * ldw_b) + sw;
    }
    const unsigned char *a_row0 = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    auto issue = [&](int kt) {
        const unsigned da = lds0 + (kt & 1) * PP::BUF + 4 * wave * 1024, dw = da + PP::OPB;
        const unsigned char *sa = a_row0 + (int64_t)kt * BKB, *sw = w_row0 + (int64_t)kt * BKB;
#pragma unroll
        for (int i = 0; i < 4; ++i) { glds16_so(voff_w[i], sw, dw + i * 1024); glds16_so(voff_a[i], sa, da + i * 1024); }
    };
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt landed (issued one whole K-tile of MFMAs ago)
        __builtin_amdgcn_s_barrier();                          // ... for every wave; and every wave is done reading tile kt-1
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) issue(kt + 1);
        const unsigned char *abuf = lds + (kt & 1) * PP::BUF + (wr * 128) * BKB;
        const unsigned char *wbuf = lds + (kt & 1) * PP::BUF + PP::OPB + (wc * 64) * BKB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 wf[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[ni] = read_frag(wbuf, ni * 16 + r, ks * 4 + q);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const uint4 af = read_frag(abuf, mi * 16 + r, ks * 4 + q);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<bf16_t>::run(wf[ni], af, acc[mi][ni]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace gemm
}  // namespace la

// ---------------------------------------------------------------------------------------------------------------------
// "mono" main loop (developer experiment, LA_PP_DBG=72): ONE wave per SIMD.  256x256 tile, 4 waves = 2 (M) x 2 (N), every
// wave a 128x128 output tile (256 accumulator VGPRs of the 512 a lone wave may use) -- a third fewer LDS fragment bytes per
// flop than the 128x64 wave tiles (16 + 16 reads per 128 MFMAs instead of 2 x (16 + 8) per 128), no partner wave: the
// wave's own stream interleaves, per k-step (K = 32, 64 MFMAs), the 16 fragment reads of the NEXT k-step (second fragment
// register set) and its 8 DMA pieces among the MFMAs.  Stages as in k2 (64-byte rows, ring of NST, one barrier per k-step).
namespace la {
namespace gemm {

struct MONO { static constexpr int THREADS = 256, NST = 4, LDS = NST * 32768; };

template <typename T16 = bf16_t>
__device__ __forceinline__ void mainloop_mono(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                              int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][8]) {
    constexpr int NST = MONO::NST, DIST = NST - 1, STAGE = 32768, OPS = 16384, SB = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int ns = K / 32;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's pieces of every stage: 4 of A (rows 64 wave .. + 63) and 4 of W
    unsigned voff_a[4], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * wave + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_a[i] = (unsigned)((int64_t)(ra - m0) * lda_b) + sw;
        voff_w[i] = (unsigned)((int64_t)(rw - n0) * ldw_b) + sw;
    }
    const unsigned char *a_row0 = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    auto issue1 = [&](int st, int slot, int i) {      // piece i (0..7: 4 of W, then 4 of A) of this wave's share of stage st
        const unsigned dst = lds0 + slot * STAGE + (i < 4 ? OPS : 0) + (4 * wave + (i & 3)) * 1024;
        if (i < 4) glds16_so(voff_w[i], w_row0 + (int64_t)st * SB, dst);
        else glds16_so(voff_a[i - 4], a_row0 + (int64_t)st * SB, dst);
    };
    auto frag_a = [&](int slot, int mi) { return read_frag2(lds + slot * STAGE + (wr * 128) * SB, mi * 16 + r, q); };
    auto frag_w = [&](int slot, int ni) { return read_frag2(lds + slot * STAGE + OPS + (wc * 128) * SB, ni * 16 + r, q); };

#pragma unroll
    for (int st = 0; st < DIST; ++st)
        if (st < ns) {
#pragma unroll
            for (int i = 0; i < 8; ++i) issue1(st, st, i);
        }
    // stage 0 landed (own pieces: all but the 8 (DIST - 1) younger ones), everyone's after the barrier; its fragments
    if (ns > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else if (ns > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
    uint4 fa[2][8], fw[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[0][i] = frag_a(0, i); fw[0][i] = frag_w(0, i); }

    int slot_n = 1, slot_w = DIST % NST;      // ring slots of stage s + 1 (next fragments) and of stage s + DIST (refill)
    auto kstep = [&](int s, auto curc) {
        constexpr int cur = decltype(curc)::value;
        // stage s + 1 must be in LDS before its fragments are read below: own pieces landed, then the barrier; the barrier
        // also says that every wave has its stage-s fragments in registers, so the slot of stage s - 1 ... wait: refill target
        // is the slot stage s - 1 occupied (read during iteration s - 2, consumed in s - 1)
        const int ahead = (ns - 1 < s + DIST - 1 ? ns - 1 : s + DIST - 1) - (s + 1);   // stages younger than s + 1 already issued
        if (s + 1 < ns) {
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        LA_PP_BARRIER();
        const bool pf = s + DIST < ns, nx = s + 1 < ns;
#pragma unroll
        for (int g = 0; g < 8; ++g) {       // 8 groups of 8 MFMAs: row tile g x all 8 column tiles
            if (nx) { fa[cur ^ 1][g] = frag_a(slot_n, g); fw[cur ^ 1][g] = frag_w(slot_n, g); }
            if (pf) issue1(s + DIST, slot_w, g);
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) Mma<T16>::run(fw[cur][ni], fa[cur][g], acc[g][ni]);
        }
        slot_n = slot_n + 1 == NST ? 0 : slot_n + 1;
        slot_w = slot_w + 1 == NST ? 0 : slot_w + 1;
    };
    for (int s = 0; s < ns; s += 2) {
        kstep(s, std::integral_constant<int, 0>{});
        if (s + 1 < ns) kstep(s + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
}

// ---------------------------------------------------------------------------------------------------------------------
// "mono", hand-placed (LA_PP_DBG=73): the same tile, stages and ring as mainloop_mono, but EVERY instruction of the k-loop is
// an `asm volatile` statement, so hipcc keeps the source order (it only allocates registers): per k-step (K = 32) the wave
// issues its 64 MFMAs back to back and drops into their gaps -- one instruction per gap, so the matrix pipe never waits for
// the issue port -- the 16 ds_read_b128 of the NEXT k-step's fragments (second register set, j = 2, 5, .., 47), its 8 LDS-DMA
// pieces of stage s + 4 (j = 4, 11, .., 53), one counted wait (j = 56) and the k-step's only barrier (j = 58).  The 8x8
// accumulators live in the 256 AGPRs.  tools/mfma_ceiling.hip runs exactly this stream on random data: 2.0 PFLOP/s.
//   Timeline (stage = k-step = 32 of K; ring slot = stage % 4; prefetch distance 4 = the whole ring):
//     k-step s computes on the fragments of stage s (in registers since k-step s - 1), reads the fragments of stage s + 1 and
//     issues the DMA of stage s + 4 into the slot of stage s itself.
//     RAW: stage s + 1 was issued during k-step s - 3; at j = 56 of k-step s - 1 every wave retires its pieces of it (the 16 of
//       stages s + 2 and s + 3 may stay in flight: vmcnt(16)), and the barrier at j = 58 publishes that.
//     WAR: the fragments of stage s were requested during k-step s - 1 (the last at j = 47) and retired by the lgkmcnt(0) at
//       j = 56 of that k-step in every wave, i.e. before the wave reached the barrier of k-step s - 1; the refill of that slot
//       is issued after it (j >= 4 of k-step s).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <typename T> struct MmaAsm;
template <> struct MmaAsm<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4 &w, const u32x4 &a, f32x4 &acc) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
    }
};
template <> struct MmaAsm<_Float16> {
    __device__ static __forceinline__ void run(const u32x4 &w, const u32x4 &a, f32x4 &acc) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
    }
};
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF> __device__ __forceinline__ void ds_read128_asm(u32x4 &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

// PROBE (timing only, wrong results): 1 = no DMA inside the loop, 2 = every refill re-reads stage 0's columns (L2-resident)
template <typename T16 = bf16_t, int PROBE = 0>
__device__ __forceinline__ void mainloop_mono_asm(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                                  int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][8]) {
    constexpr int STAGE = 32768, OPS = 16384, SB = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int ns = K / 32;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's pieces of every stage: 4 of W (rows 64 wave .. + 63) and 4 of A
    unsigned voff_a[4], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * wave + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int ra = m0 + rt; ra = ra > M - 1 ? M - 1 : ra;
        int rw = n0 + rt; rw = rw > N - 1 ? N - 1 : rw;
        voff_a[i] = (unsigned)((int64_t)(ra - m0) * lda_b) + sw;
        voff_w[i] = (unsigned)((int64_t)(rw - n0) * ldw_b) + sw;
    }
    const unsigned char *a_row0 = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_row0 = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    const unsigned piece0 = lds0 + (4 * wave) * 1024;
    // piece i (0..3 of W, 4..7 of A) of this wave's share of the stage whose K offset is kb bytes, into ring slot `slot`
    auto issue1 = [&](const unsigned char *w_src, const unsigned char *a_src, int slot, int i) __attribute__((always_inline)) {
        const unsigned dst = piece0 + slot * STAGE + (i < 4 ? OPS : 0) + (i & 3) * 1024;
        if (i < 4) glds16_so(voff_w[i], w_src, dst);
        else glds16_so(voff_a[i - 4], a_src, dst);
    };
    // fragment addresses: lane part + slot * 32 KiB + tile * 1 KiB (swz2 does not depend on the 16-row tile index); the
    // ds_read offset field holds 16 bits, so slots 2 and 3 go through a second base register
    const unsigned fa_lo = lds0 + (unsigned)((wr * 128 + r) * SB + ((q ^ swz2(r)) << 4)), fa_hi = fa_lo + 2 * STAGE;
    const unsigned fw_lo = lds0 + OPS + (unsigned)((wc * 128 + r) * SB + ((q ^ swz2(r)) << 4)), fw_hi = fw_lo + 2 * STAGE;

#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int i = 0; i < 8; ++i) issue1(w_row0 + st * SB, a_row0 + st * SB, st, i);
    }
    // stages 0 and 1 landed (own pieces; everyone's after the barrier): the 16 pieces of stages 2 and 3 may stay in flight
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    LA_PP_BARRIER();
    u32x4 fa[2][8], fw[2][8];
    ds_read128_asm<0 * 1024>(fw[0][0], fw_lo); ds_read128_asm<1 * 1024>(fw[0][1], fw_lo); ds_read128_asm<2 * 1024>(fw[0][2], fw_lo);
    ds_read128_asm<3 * 1024>(fw[0][3], fw_lo); ds_read128_asm<4 * 1024>(fw[0][4], fw_lo); ds_read128_asm<5 * 1024>(fw[0][5], fw_lo);
    ds_read128_asm<6 * 1024>(fw[0][6], fw_lo); ds_read128_asm<7 * 1024>(fw[0][7], fw_lo);
    ds_read128_asm<0 * 1024>(fa[0][0], fa_lo); ds_read128_asm<1 * 1024>(fa[0][1], fa_lo); ds_read128_asm<2 * 1024>(fa[0][2], fa_lo);
    ds_read128_asm<3 * 1024>(fa[0][3], fa_lo); ds_read128_asm<4 * 1024>(fa[0][4], fa_lo); ds_read128_asm<5 * 1024>(fa[0][5], fa_lo);
    ds_read128_asm<6 * 1024>(fa[0][6], fa_lo); ds_read128_asm<7 * 1024>(fa[0][7], fa_lo);
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[1][i] = fa[0][i]; fw[1][i] = fw[0][i]; }     // (defined values for the never-used first set)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // slot 0 is refilled in k-step 0: every wave's stage-0 fragments
    LA_PP_BARRIER();                                       // are in registers first

    // one k-step.  NX: there is a stage s + 1 (its fragments are read), PF: there is a stage s + 4 (its DMA is issued), VM: the
    // pieces this wave may leave in flight at the k-step's vmcnt -- all compile-time: the steady loop has (1, 1, 16), the four
    // peeled last k-steps (1,0,8), (1,0,0), (1,0,0), (0,0,0); no branch anywhere.
    auto kstep = [&](int s, auto nxc, auto pfc, auto vmc, auto curc, auto slotc) __attribute__((always_inline)) {
        constexpr bool NX = decltype(nxc)::value, PF = decltype(pfc)::value && PROBE != 1;
        constexpr int VM = decltype(vmc)::value;
        constexpr int CUR = decltype(curc)::value, SLOT = decltype(slotc)::value;
        constexpr int SN = (SLOT + 1) & 3, SW = SLOT;            // stage s + 4 takes the slot of stage s (free since barrier s - 1)
        constexpr int OFFN = (SN & 1) * STAGE;                   // offset of slot SN from its base register
        const unsigned fan = SN >= 2 ? fa_hi : fa_lo, fwn = SN >= 2 ? fw_hi : fw_lo;
        const unsigned char *w_src = w_row0 + (PROBE == 2 ? 0 : (int64_t)(s + 4) * SB), *a_src = a_row0 + (PROBE == 2 ? 0 : (int64_t)(s + 4) * SB);
        static_for<0, 64>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, mi = j >> 3, ni = j & 7;
            MmaAsm<T16>::run(fw[CUR][ni], fa[CUR][mi], acc[mi][ni]);
            if constexpr (NX && j % 3 == 2 && j / 3 < 16) {
                constexpr int i = j / 3;                   // fragments of the next k-step: W 0..7, then A 0..7
                if constexpr (i < 8) ds_read128_asm<OFFN + i * 1024>(fw[CUR ^ 1][i], fwn);
                else ds_read128_asm<OFFN + (i - 8) * 1024>(fa[CUR ^ 1][i - 8], fan);
            }
            if constexpr (PF && j % 7 == 4 && j / 7 < 8) issue1(w_src, a_src, SW, j / 7);
            if constexpr (j == 56) {
                // stage s + 2 (read during k-step s + 1) has landed; the next stage's fragments (last requested 9 MFMAs ago)
                // are in registers, so its slot may be refilled after the barrier
                if constexpr (VM == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
                else if constexpr (VM == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            if constexpr (j == 58) asm volatile("s_barrier" ::: "memory");
        });
    };
    typedef std::false_type F;
    typedef std::true_type TT;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    typedef std::integral_constant<int, 8> V8;
    typedef std::integral_constant<int, 16> V16;
    int s = 0;                              // ns is a multiple of 4, >= 8 (host check)
    for (; s + 4 < ns; s += 4) {            // every k-step here has a stage s + 4
        kstep(s, TT{}, TT{}, V16{}, I0{}, I0{});
        kstep(s + 1, TT{}, TT{}, V16{}, I1{}, I1{});
        kstep(s + 2, TT{}, TT{}, V16{}, I0{}, I2{});
        kstep(s + 3, TT{}, TT{}, V16{}, I1{}, I3{});
    }
    kstep(s, TT{}, F{}, V8{}, I0{}, I0{});
    kstep(s + 1, TT{}, F{}, I0{}, I1{}, I1{});
    kstep(s + 2, TT{}, F{}, I0{}, I0{}, I2{});
    kstep(s + 3, F{}, F{}, I0{}, I1{}, I3{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
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

// PLACE (compile-time A/B, measured once and left in the source): where the reads / DMA pieces / wait / barrier sit among the
// 32 MFMAs.  All four land within +-1 % of each other on the encoder's shapes -- the loop is LDS-bandwidth-bound (96 KiB of
// fragment reads + 32 KiB of DMA writes per k-step = the 128 B per clock of a 1024-cycle k-step), not placement-bound.
//   0: reads j = 1, 3, .., 23; DMA j = 2, 10, 18, 26; wait 28; barrier 29 (shipped)
//   1: reads j = 0 .. 11 (one per MFMA, early); DMA j = 13, 17, 21, 25; wait 28; barrier 29
//   2: reads j = 1, 3, .., 23; DMA j = 0, 2, 4, 6 (early burst); wait 28; barrier 29
//   3: as 0 with the wait at 30 and the barrier at 31 (end of the k-step)
//   4: as 0 for the waves wr = 0; their SIMD partners (wr = 1) read at even j and issue their DMA pieces at j = 5, 13, 21, 26
//      (so that the two waves of a SIMD are not held at an LDS-DMA issue together): 2.5 % SLOWER on the K = 1024 shapes
template <typename T16 = bf16_t, int PLACE = 0>
__device__ __forceinline__ void mainloop_duo_asm(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                                 int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4]) {
    constexpr int STAGE = 32768, OPS = 16384, SB = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int ns = K / 32;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's 4 pieces of every stage: waves 0..3 carry the W image (rows 64 wave .. + 63), waves 4..7 the A image
    const bool is_w = wave < 4;
    const int pw = wave & 3;
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * pw + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int rg = (is_w ? n0 : m0) + rt;
        const int lim = (is_w ? N : M) - 1;
        rg = rg > lim ? lim : rg;
        voff[i] = (unsigned)((int64_t)(rg - (is_w ? n0 : m0)) * (is_w ? ldw_b : lda_b)) + sw;
    }
    const unsigned char *src0 = is_w ? reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b
                                     : reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    const unsigned piece0 = lds0 + (is_w ? OPS : 0) + (4 * pw) * 1024;
    auto issue1 = [&](const unsigned char *src, int slot, int i) __attribute__((always_inline)) {
        glds16_so(voff[i], src, piece0 + slot * STAGE + i * 1024);
    };
    const unsigned fa_lo = lds0 + (unsigned)((wr * 128 + r) * SB + ((q ^ swz2(r)) << 4)), fa_hi = fa_lo + 2 * STAGE;
    const unsigned fw_lo = lds0 + OPS + (unsigned)((wc * 64 + r) * SB + ((q ^ swz2(r)) << 4)), fw_hi = fw_lo + 2 * STAGE;

#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue1(src0 + st * SB, st, i);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // stages 0 and 1 landed; the 8 pieces of stages 2 and 3 may stay in flight
    LA_PP_BARRIER();
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

    // STAG (PLACE == 4): the two waves of a SIMD (wave w and w + 4, i.e. wr = 0 / 1) carry their reads and DMA pieces in
    // different MFMA gaps -- wr = 1 has its reads at even j and its DMA pieces four MFMAs later -- so that they are not
    // both held at an LDS-DMA issue at the same moment
    auto kstep = [&](int s, auto nxc, auto pfc, auto vmc, auto curc, auto slotc, auto stagc) __attribute__((always_inline)) {
        constexpr bool NX = decltype(nxc)::value, PF = decltype(pfc)::value;
        constexpr int VM = decltype(vmc)::value;
        constexpr int CUR = decltype(curc)::value, SLOT = decltype(slotc)::value;
        constexpr bool STAG = decltype(stagc)::value;
        constexpr int SN = (SLOT + 1) & 3, SW = SLOT;
        constexpr int OFFN = (SN & 1) * STAGE;
        const unsigned fan = SN >= 2 ? fa_hi : fa_lo, fwn = SN >= 2 ? fw_hi : fw_lo;
        const unsigned char *src = src0 + (int64_t)(s + 4) * SB;
        static_for<0, 32>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, mi = j >> 2, ni = j & 3;
            MmaAsmV<T16>::run(fw[CUR][ni], fa[CUR][mi], acc[mi][ni]);
            constexpr bool rd_here = PLACE == 1 ? j < 12 : STAG ? ((j & 1) == 0 && j / 2 < 12) : ((j & 1) == 1 && j / 2 < 12);
            if constexpr (NX && rd_here) {
                constexpr int i = PLACE == 1 ? j : j / 2;  // fragments of the next k-step: W 0..3, then A 0..7
                if constexpr (i < 4) ds_read128_asm<OFFN + i * 1024>(fw[CUR ^ 1][i], fwn);
                else ds_read128_asm<OFFN + (i - 4) * 1024>(fa[CUR ^ 1][i - 4], fan);
            }
            constexpr bool dma_here = PLACE == 1 ? (j >= 13 && j <= 25 && (j - 13) % 4 == 0) : PLACE == 2 ? (j < 8 && (j & 1) == 0)
                                      : STAG ? (j == 5 || j == 13 || j == 21 || j == 26) : (j & 7) == 2;
            constexpr int dma_i = PLACE == 1 ? (j - 13) / 4 : PLACE == 2 ? j / 2 : (STAG && j == 26) ? 3 : j >> 3;
            if constexpr (PF && dma_here) issue1(src, SW, dma_i);
            if constexpr (j == (PLACE == 3 ? 30 : 28)) {
                if constexpr (VM == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else if constexpr (VM == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            if constexpr (j == (PLACE == 3 ? 31 : 29)) asm volatile("s_barrier" ::: "memory");
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
    auto run = [&](auto stagc) __attribute__((always_inline)) {
        int s = 0;                              // ns is a multiple of 4, >= 8 (host check)
        for (; s + 4 < ns; s += 4) {
            kstep(s, TT{}, TT{}, V8{}, I0{}, I0{}, stagc);
            kstep(s + 1, TT{}, TT{}, V8{}, I1{}, I1{}, stagc);
            kstep(s + 2, TT{}, TT{}, V8{}, I0{}, I2{}, stagc);
            kstep(s + 3, TT{}, TT{}, V8{}, I1{}, I3{}, stagc);
        }
        kstep(s, TT{}, F{}, V4{}, I0{}, I0{}, stagc);
        kstep(s + 1, TT{}, F{}, I0{}, I1{}, I1{}, stagc);
        kstep(s + 2, TT{}, F{}, I0{}, I0{}, I2{}, stagc);
        kstep(s + 3, F{}, F{}, I0{}, I1{}, I3{}, stagc);
    };
    if constexpr (PLACE == 4) {
        if (wr == 0) run(F{}); else run(TT{});          // wave-uniform: both arms execute the same number of barriers
    } else {
        run(F{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
}

}  // namespace gemm
}  // namespace la
