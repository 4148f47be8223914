// la_gemm_lab_loops.h -- EXPERIMENT build only (-DLA_EXPERIMENTS, tools/build_variant.sh): main loops of the GEMM structures that were
// built, held bit-identical to the shipped kernel and MEASURED SLOWER in rounds 2-4 (profiles/NOTES.md): the one-wave-per-SIMD
// "mono" tile and the four-wave, two-workgroups-per-CU "q4" form.  Not part of liblyricalign_hip.so's default build.
#pragma once
#include "../la_gemm_pp.h"

// ---------------------------------------------------------------------------------------------------------------------
// "mono" tile (LA_PP_DBG=73): ONE wave per SIMD.  256x256 tile, 4 waves = 2 (M) x 2 (N), every wave a 128x128 output tile
// (256 accumulator registers of the 512 a lone wave may use) -- a third fewer LDS fragment bytes per flop than the 128x64
// wave tiles (16 + 16 reads per 128 MFMAs instead of 2 x (16 + 8) per 128), no partner wave: the wave's own stream
// interleaves, per k-step (K = 32, 64 MFMAs), the 16 fragment reads of the NEXT k-step (second fragment register set) and
// its 8 DMA pieces among the MFMAs.  64-byte-row stages, ring of 4, one barrier per k-step.
namespace la {
namespace gemm {

struct MONO { static constexpr int THREADS = 256, NST = 4, LDS = NST * 32768; };

// ---------------------------------------------------------------------------------------------------------------------
// The hand-placed stream of the mono tile: EVERY instruction of the k-loop is
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

template <typename T16 = bf16_t>
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
        constexpr bool NX = decltype(nxc)::value, PF = decltype(pfc)::value;
        constexpr int VM = decltype(vmc)::value;
        constexpr int CUR = decltype(curc)::value, SLOT = decltype(slotc)::value;
        constexpr int SN = (SLOT + 1) & 3, SW = SLOT;            // stage s + 4 takes the slot of stage s (free since barrier s - 1)
        constexpr int OFFN = (SN & 1) * STAGE;                   // offset of slot SN from its base register
        const unsigned fan = SN >= 2 ? fa_hi : fa_lo, fwn = SN >= 2 ? fw_hi : fw_lo;
        const unsigned char *w_src = w_row0 + (int64_t)(s + 4) * SB, *a_src = a_row0 + (int64_t)(s + 4) * SB;
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
// "q4" (round 4, verdict item 2b; LA_GEMM_Q4=1): FOUR-wave workgroups, TWO of them resident per CU.  A workgroup owns a 256 x 128
// tile -- 4 waves = 2 (M) x 2 (N), the same 128x64 wave tiles, accumulator layout and epilogues as the 8-wave kernel -- with its own
// ring of THREE K = 32 stages of 24 KiB (A 256 rows x 64 B, then W 128 rows x 64 B; 72 KiB + 2 KiB of row statistics per
// workgroup, 148 KiB per CU) and one wave on every SIMD.  The two workgroups of a CU are independent tiles that the hardware
// interleaves: one's prologue, epilogue and dispatch gap run under the other's main loop (tools/tile_timeline.py: 26-43 % of a
// K = 1024 tile's life at one workgroup per CU).  Price: the W stage is staged per 128 columns instead of per 256 (48 instead of
// 32 KiB of LDS-DMA per 256 x 256 x 32 of product), and the prefetch distance is 3 stages instead of 4.
//   Per k-step s and wave: 32 MFMAs on fragment set s & 1, the 12 fragment reads of stage s + 1 (slot (s + 1) % 3), the wave's 6
//   DMA pieces of stage s + 3 into slot s % 3 (4 of A, 2 of W; that slot's fragments were read during k-step s - 1), one counted
//   wait (vmcnt(6): stage s + 2 has landed, stage s + 3 may stay in flight) and one barrier.  Fragment sets alternate with period 2,
//   slots with period 3: the loop body is six k-steps, the tail (R = 4, 6 or 8 k-steps; K is a multiple of 128) is spelled out.
struct Q4 {
    static constexpr int THREADS = 256, TM = 256, TN = 128, NST = 3, STAGE = 24576, OPS_W = 16384, STATS = NST * STAGE, LDS = NST * STAGE + 2048;
    // k-steps left after the six-step loop body has run while nine or more remain (K / 32 = ns a multiple of 4, >= 8): 4, 6 or 8
    static constexpr int rem_of(int ns) { return ns - 6 * ((ns - 3) / 6); }
};

template <typename T16 = bf16_t, int REM = 8, bool WIDE = false>
__device__ __forceinline__ void mainloop_q4_asm(const T16 *A, int64_t lda, int M, const T16 *W, int64_t ldw, int N, int K,
                                                int m0, int n0, unsigned char *lds, f32x4 (&acc)[8][4] LA_STAMP_PARAM) {
    // WIDE = false: 256 x 128 tiles, waves 2 (M) x 2 (N), stage = A 16 KiB then W 8 KiB; WIDE = true: 128 x 256 tiles, waves 1 x 4,
    // stage = W 16 KiB then A 8 KiB (the A panel -- the operand that streams from HBM -- is then staged once per 256 columns, as in
    // the 8-wave kernel; the W panel, which lives in L2, twice as often).  "big" = the 256-row operand: 4 pieces per wave and stage.
    constexpr int STAGE = Q4::STAGE, OPS = Q4::OPS_W, SB = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_u32(lds));
    const int wr = WIDE ? 0 : wave >> 1, wc = WIDE ? wave : wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int64_t lda_b = lda * 2, ldw_b = ldw * 2;
    const int ns = K / 32;
    const int big0 = WIDE ? n0 : m0, small0 = WIDE ? m0 : n0, big_lim = (WIDE ? N : M) - 1, small_lim = (WIDE ? M : N) - 1;
    const int64_t big_ld = WIDE ? ldw_b : lda_b, small_ld = WIDE ? lda_b : ldw_b;
    unsigned voff[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rt = (4 * wave + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int rg = big0 + rt;
        rg = rg > big_lim ? big_lim : rg;
        voff[i] = (unsigned)((int64_t)(rg - big0) * big_ld) + sw;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = (2 * wave + i) * 16 + (lane >> 2);
        const int sw = ((lane & 3) ^ swz2(rt)) << 4;
        int rg = small0 + rt;
        rg = rg > small_lim ? small_lim : rg;
        voff[4 + i] = (unsigned)((int64_t)(rg - small0) * small_ld) + sw;
    }
    const unsigned char *a_rows = reinterpret_cast<const unsigned char *>(A) + (int64_t)m0 * lda_b;
    const unsigned char *w_rows = reinterpret_cast<const unsigned char *>(W) + (int64_t)n0 * ldw_b;
    const unsigned char *srcA = WIDE ? w_rows : a_rows;           // (srcA / pieceA: the 256-row operand; srcW / pieceW: the 128-row one)
    const unsigned char *srcW = WIDE ? a_rows : w_rows;
    const unsigned pieceA = lds0 + (4 * wave) * 1024, pieceW = lds0 + OPS + (2 * wave) * 1024;
    const unsigned fa = lds0 + (WIDE ? OPS : 0) + (unsigned)((wr * 128 + r) * SB + ((q ^ swz2(r)) << 4));
    const unsigned fw = lds0 + (WIDE ? 0 : OPS) + (unsigned)((wc * 64 + r) * SB + ((q ^ swz2(r)) << 4));
    auto issue1 = [&](int st, int slot, int i) __attribute__((always_inline)) {
        if (i < 4) glds16_so(voff[i], srcA + (int64_t)st * SB, pieceA + slot * STAGE + i * 1024);
        else glds16_so(voff[i], srcW + (int64_t)st * SB, pieceW + slot * STAGE + (i - 4) * 1024);
    };
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 3; ++st) {
#pragma unroll
        for (int i = 0; i < 6; ++i) issue1(st, st, i);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // stages 0 and 1 landed; stage 2's 6 pieces may stay in flight
    LA_PP_BARRIER();
    LA_STAMP_T1();
    u32x4 fa_[2][8], fw_[2][4];
    ds_read128_asm<0 * 1024>(fw_[0][0], fw); ds_read128_asm<1 * 1024>(fw_[0][1], fw); ds_read128_asm<2 * 1024>(fw_[0][2], fw);
    ds_read128_asm<3 * 1024>(fw_[0][3], fw);
    ds_read128_asm<0 * 1024>(fa_[0][0], fa); ds_read128_asm<1 * 1024>(fa_[0][1], fa); ds_read128_asm<2 * 1024>(fa_[0][2], fa);
    ds_read128_asm<3 * 1024>(fa_[0][3], fa); ds_read128_asm<4 * 1024>(fa_[0][4], fa); ds_read128_asm<5 * 1024>(fa_[0][5], fa);
    ds_read128_asm<6 * 1024>(fa_[0][6], fa); ds_read128_asm<7 * 1024>(fa_[0][7], fa);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa_[1][i] = fa_[0][i];
#pragma unroll
    for (int i = 0; i < 4; ++i) fw_[1][i] = fw_[0][i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
    __builtin_amdgcn_s_setprio(0);                              // (the kernel may have raised it for its prologue: LA_GEMM_Q4_PRIO)

    // KIND: 0 = full (DMA of stage s + 3, vmcnt(6)), 1 = tail (no DMA left to issue: vmcnt(0)), 2 = last (no next fragments either)
    auto kstep = [&](int s, auto phc, auto kindc) __attribute__((always_inline)) {
        constexpr int PH = decltype(phc)::value, KIND = decltype(kindc)::value;
        constexpr int CUR = PH & 1, SLOT = PH % 3, SN = (SLOT + 1) % 3;
        constexpr int OFFN = SN * STAGE;
        static_for<0, 32>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, mi = j >> 2, ni = j & 3;
            MmaAsmV<T16>::run(fw_[CUR][ni], fa_[CUR][mi], acc[mi][ni]);
            if constexpr (KIND != 2 && (j & 1) == 1 && j / 2 < 12) {
                constexpr int i = j / 2;                        // fragments of the next k-step: W 0..3, then A 0..7
                if constexpr (i < 4) ds_read128_asm<OFFN + i * 1024>(fw_[CUR ^ 1][i], fw);
                else ds_read128_asm<OFFN + (i - 4) * 1024>(fa_[CUR ^ 1][i - 4], fa);
            }
            if constexpr (KIND == 0 && (j & 3) == 2 && j < 24) issue1(s + 3, SLOT, j >> 2);
            if constexpr (j == 28) {
                if constexpr (KIND == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            if constexpr (j == 29) asm volatile("s_barrier" ::: "memory");
        });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    typedef std::integral_constant<int, 4> I4;
    typedef std::integral_constant<int, 5> I5;
    int s = 0;                                                  // ns is a multiple of 4, >= 8 (host check)
    for (; s + 9 <= ns; s += 6) {
        kstep(s, I0{}, I0{}); kstep(s + 1, I1{}, I0{}); kstep(s + 2, I2{}, I0{});
        kstep(s + 3, I3{}, I0{}); kstep(s + 4, I4{}, I0{}); kstep(s + 5, I5{}, I0{});
    }
    // REM = 4, 6 or 8 k-steps are left (chosen by the HOST from K: three alternative tails in one kernel made hipcc spill ~1600
    // registers), of which REM - 3 still issue a stage
    if constexpr (REM == 8) {
        kstep(s, I0{}, I0{}); kstep(s + 1, I1{}, I0{}); kstep(s + 2, I2{}, I0{}); kstep(s + 3, I3{}, I0{}); kstep(s + 4, I4{}, I0{});
        kstep(s + 5, I5{}, I1{}); kstep(s + 6, I0{}, I1{}); kstep(s + 7, I1{}, I2{});
    } else if constexpr (REM == 6) {
        kstep(s, I0{}, I0{}); kstep(s + 1, I1{}, I0{}); kstep(s + 2, I2{}, I0{});
        kstep(s + 3, I3{}, I1{}); kstep(s + 4, I4{}, I1{}); kstep(s + 5, I5{}, I2{});
    } else {
        kstep(s, I0{}, I0{});
        kstep(s + 1, I1{}, I1{}); kstep(s + 2, I2{}, I1{}); kstep(s + 3, I3{}, I2{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LA_PP_BARRIER();
}

}  // namespace gemm
}  // namespace la
