// la_gemm_lab.hip -- EXPERIMENT build only (tools/build_variant.sh compiles this file with -DLA_EXPERIMENTS; the default build of
// liblyricalign_hip.so never sees it).  The GEMM structures of rounds 2-4 that were built, held bit-identical to the shipped 256 x 256
// kernel and MEASURED SLOWER (profiles/NOTES.md), behind their per-launch developer switches:
//   LA_GEMM_Q4=1|2       gemm_q4_kernel: four-wave workgroups, two resident per CU (256 x 128 / 128 x 256 tiles)
//   LA_GEMM_PERSIST=1    gemm_pp_persist_kernel: persistent workgroups drawing tiles from per-XCD tickets, next tile's stages prefetched
//   LA_PP_DBG=73         gemm_mono_kernel: one wave per SIMD, 128 x 128 wave tiles, accumulators in AGPRs
//   ln_csum w/o ln_stats gemm_pp_kernel<.., LNM = 4>: LayerNorm row statistics taken by the consumer's main loop (v_dot2c in MFMA gaps)
#ifndef LA_EXPERIMENTS
#error "lab/la_gemm_lab.hip belongs to the experiment build (-DLA_EXPERIMENTS, tools/build_variant.sh)"
#endif
#include <mutex>
#include <vector>

#include "../la_gemm_pp_kernel.h"
#include "la_gemm_lab_loops.h"

namespace la {
namespace gemm {

// ---------------------------------------------------------------------------------------------------------------------
// Four-wave workgroups, two resident per CU (mainloop_q4_asm, la_gemm_pp.h; LA_GEMM_Q4=1): 256 x 128 tiles, the 8-wave kernel's wave
// tiles and epilogues.  One workgroup's prologue / epilogue / dispatch gap runs under the other's main loop.
template <bool OUT_F32, typename T16, int LNM = 0, int REM = 8, bool WIDE = false>
__global__ __launch_bounds__(Q4::THREADS, 2) void gemm_q4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_t1 = stamp_t0;
#endif
    // LA_GEMM_Q4_PRIO=1: prologue and epilogue at raised wave priority -- they are short, latency-bound phases that share the SIMDs
    // with the OTHER workgroup's main loop; at equal priority the timeline shows them stretched 2-4 x (tools/tile_timeline.py).
    const bool prio = p.epilogue & LA_EPI_Q4_PRIO;
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord_mb(tile, p.tiles_m, p.tiles_n, p.group, p.mblock);
    constexpr int TM = WIDE ? Q4::TN : Q4::TM, TN = WIDE ? Q4::TM : Q4::TN;      // 256 x 128, or 128 x 256 (WIDE)
    const int m0 = tc.tm * TM, n0 = tc.tn * TN;
    const int z = blockIdx.y;
    const T16 *A = reinterpret_cast<const T16 *>(p.A) + (int64_t)z * p.strideA;
    const T16 *W = reinterpret_cast<const T16 *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = WIDE ? 0 : wave >> 1, wc = WIDE ? wave : wave & 1;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    const int ncol = min(n0 + wc * 64 + lane, p.N - 1);          // per-column operands before the main loop (as gemm_pp_kernel)
    const float bias_l = has_bias ? bias[ncol] : 0.f;
    float csum_l = 0.f;
    if constexpr (LNM == 2) csum_l = p.ln_csum[ncol];
    float2 st_pre = make_float2(0.f, 1.f);
    constexpr int SROWS = TM / 4;                               // rows of the tile whose statistics this wave fetches
    if constexpr (LNM == 2) {
        if (lane < SROWS) st_pre = reinterpret_cast<const float2 *>(p.ln_stats)[min(m0 + wave * SROWS + lane, p.M - 1)];
    }

    f32x4 acc[8][4];
    mainloop_q4_asm<T16, REM, WIDE>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG);
    if (prio) __builtin_amdgcn_s_setprio(3);
    float2 *stats_tab = reinterpret_cast<float2 *>(lds + Q4::STATS);         // behind the ring: clear of the epilogue staging
    if constexpr (LNM == 2) {
        if (lane < SROWS) stats_tab[wave * SROWS + lane] = st_pre;
    }
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
    if constexpr (LNM == 3) {
        wave_epilogue_split<T16, 1>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, lds + wave * (32 * EPI_PITCH), lds + wave * (32 * EPI_PITCH));
    } else wave_epilogue<OUT_F32, T16, LNM == 2 ? 4 : LNM>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, csum_l, lds + wave * (32 * EPI_PITCH), stats_tab, m0);
#ifdef LA_TILE_STAMPS
    if (threadIdx.x == 0 && g_tile_stamps) {
        unsigned long long *o = g_tile_stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = stamp_t0; o[1] = stamp_t1; o[2] = stamp_t2; o[3] = __builtin_amdgcn_s_memrealtime();
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        o[4] = hw_id; o[5] = xcc_id;
        o[6] = tile;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the 256x256 kernel (round 4; hand-placed main loop, batch 1, 16-bit operands).
// tools/tile_timeline.py: 26-43 % of a K = 1024 tile's time on its CU is not the main loop -- 2.4-3.0 us of prologue (waiting for its
// first two stages), the epilogue, and 1.1-4.0 us until the next workgroup enters (a wave retires only when its stores are
// acknowledged).  Here a workgroup stays: it DRAWS tiles from a ticket counter (one per XCD, so an XCD keeps walking its own
// contiguous run of tile ids -- the L2 sharing of the hardware-dispatched form -- and steals from the next XCD's run when its own
// is empty), and before it starts a tile's epilogue it has already issued the NEXT tile's stages 0 .. 3 into the ring: they land
// under the epilogue, the epilogue's stores drain under the next main loop (duo_run PREFETCHED: its first waits are counted
// so that no store is forced), and nothing waits for a dispatch.  Dynamic tickets keep what the dispatcher gives for free: a
// workgroup that starts late (CUs held by the head stream's resident recurrence) simply draws fewer tiles.
//   * Ticket: wave 0 issues a returning atomic at the TOP of a tile's main loop and reads it at the end (in flight ~1 us, the loop
//     takes 25-100).  The result lands asynchronously, so it is parked in a register hipcc does not manage across that span: the
//     physical v255, named in both asm statements and as their clobber (the kernel's other values sit in v0 .. ~v240; the build
//     checks the assembly: v255 may not be written between the two).  An AGPR would be the natural home, but any AGPR use makes
//     hipcc split the 256-register budget 128 / 128 and spill ~550 registers.  Broadcast through one LDS word, two barriers.
//   * LDS: all 160 KiB -- ring slots [0, 128 K) and 8 x 4 KiB of epilogue staging behind them (wave_epilogue STG = 1); the LayerNorm
//     statistics travel in registers (LNM = 5).  A partial (edge) tile takes the element-wise epilogue with its 32-row staging at the
//     ring's front, so nothing is prefetched before it.
//   * Exit: the last workgroup to leave (a second counter) zeroes the tickets for the next launch on the stream.
struct PersistTickets { unsigned head[8]; unsigned done; unsigned pad[7]; };

template <bool OUT_F32, typename T16, int LNM>
__global__ __launch_bounds__(PP::THREADS, 2) void gemm_pp_persist_kernel(GemmParams p, PersistTickets *tk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int EM = LNM == 2 ? 5 : LNM;                   // epilogue form: LayerNorm statistics from the wave's registers
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(la::lds_addr_u32(lds));
    const int nt = p.tiles_m * p.tiles_n;
    const int q8 = nt >> 3, r8 = nt & 7;
    unsigned xcc_raw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_raw));
    int cur_x = (int)(xcc_raw & 7u);                         // the XCD whose run of tile ids this workgroup currently draws from
    int tried = 0;                                           // runs found empty so far
    int *tslot = reinterpret_cast<int *>(lds);               // ticket broadcast word (ring slot 0; free whenever it is used)
    const T16 *A = reinterpret_cast<const T16 *>(p.A);
    const T16 *W = reinterpret_cast<const T16 *>(p.W);
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && p.bias;

    // ticket k of run x -> tile id, or -1 and move on to the next run (wave 0 only; all values wave-uniform)
    auto resolve = [&](unsigned k) -> int {
        while (true) {
            const int len = q8 + (cur_x < r8 ? 1 : 0);
            if ((int)k < len) return (cur_x < r8 ? cur_x * (q8 + 1) : r8 * (q8 + 1) + (cur_x - r8) * q8) + (int)k;
            if (++tried >= 8) return -1;
            cur_x = (cur_x + 1) & 7;
            unsigned kk = 0;
            if ((tid & 63) == 0) kk = __hip_atomic_fetch_add(&tk->head[cur_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            k = __builtin_amdgcn_readfirstlane(kk);
        }
    };
    auto broadcast = [&](int v) -> int {                     // wave 0's value to every wave (the ring is free at both call sites)
        if (tid == 0) *tslot = v;
        __syncthreads();
        const int r = __builtin_amdgcn_readfirstlane(*tslot);
        __syncthreads();
        return r;
    };

    int t = 0;
    if (wave == 0) {
        unsigned kk = 0;
        if ((tid & 63) == 0) kk = __hip_atomic_fetch_add(&tk->head[cur_x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = resolve(__builtin_amdgcn_readfirstlane(kk));
    }
    t = broadcast(t);

    la::gemm::DuoCtx c;
    float bias_l = 0.f, csum_l = 0.f;
    float2 sr0 = make_float2(0.f, 1.f), sr1 = make_float2(0.f, 1.f);
    int m0 = 0, n0 = 0;
    // everything a tile needs before its main loop: coordinates, DMA addresses, the stages 0 .. 3, the epilogue's per-column /
    // per-row operands (requested here, used after the loop)
    auto open_tile = [&](int tile, la::gemm::DuoCtx &cc, int &mm0, int &nn0, float &b_l, float &cs_l, float2 &s0, float2 &s1) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));                       // (keeps hipcc from hoisting a tile's lane arithmetic across the other tile's loop)
        const TileCoord tc = tile_coord_mb(tile, p.tiles_m, p.tiles_n, p.group, p.mblock);
        mm0 = tc.tm * PP::TM; nn0 = tc.tn * PP::TN;
        la::gemm::duo_setup<T16>(cc, A, p.lda, p.M, W, p.ldw, p.N, mm0, nn0, lds0, wave, lane);
        la::gemm::duo_issue_prologue(cc);
        const int ncol = min(nn0 + wc * 64 + lane, p.N - 1);
        b_l = has_bias ? p.bias[ncol] : 0.f;
        if constexpr (LNM == 2) {
            cs_l = p.ln_csum[ncol];
            const float2 *st = reinterpret_cast<const float2 *>(p.ln_stats);
            s0 = st[min(mm0 + wr * 128 + lane, p.M - 1)];
            s1 = st[min(mm0 + wr * 128 + 64 + lane, p.M - 1)];
        }
    };
    if (t >= 0) open_tile(t, c, m0, n0, bias_l, csum_l, sr0, sr1);
    bool prefetched = false;
    for (int guard = 0; t >= 0 && guard <= nt; ++guard) {
        if (wave == 0 && tried < 8) {                        // the next ticket: in flight during the main loop, parked in v255
            unsigned one = 1;
            unsigned *hp = &tk->head[cur_x];
            if ((tid & 63) == 0)
                asm volatile("global_atomic_add v255, %0, %1, off sc0" ::"v"(hp), "v"(one) : "memory", "v255");
        }
        f32x4 acc[8][4];
#ifdef LA_TILE_STAMPS
        const unsigned long long ps0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long ps1 = ps0;
        if (prefetched) la::gemm::duo_run<T16, -1, true>(c, p.K, acc, ps1);
        else la::gemm::duo_run<T16, -1, false>(c, p.K, acc, ps1);
        const unsigned long long ps2 = __builtin_amdgcn_s_memrealtime();
#else
        if (prefetched) la::gemm::duo_run<T16, -1, true>(c, p.K, acc);
        else la::gemm::duo_run<T16, -1, false>(c, p.K, acc);
#endif
        int tn = -1;
        if (wave == 0 && tried < 8) {
            unsigned kk = 0;
            asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v255" : "=v"(kk)::"memory", "v255");
            tn = resolve(__builtin_amdgcn_readfirstlane(kk));
        }
        tn = broadcast(tn);
        const bool interior = m0 + PP::TM <= p.M && n0 + PP::TN <= p.N;
        const bool chain = tn >= 0 && interior;
        la::gemm::DuoCtx cn;
        float bias_n = 0.f, csum_n = 0.f;
        float2 sn0 = make_float2(0.f, 1.f), sn1 = make_float2(0.f, 1.f);
        int mn0 = 0, nn0 = 0;
        if (chain) open_tile(tn, cn, mn0, nn0, bias_n, csum_n, sn0, sn1);      // its stages land under this tile's epilogue
        unsigned char *stg = lds + 4 * 32768 + wave * 4096, *stg_edge = lds + wave * (32 * EPI_PITCH);
#ifdef LA_TILE_STAMPS
        const unsigned long long ps3 = __builtin_amdgcn_s_memrealtime();
#endif
        int lane_e = tid & 63;
        asm volatile("" : "+v"(lane_e));
        if constexpr (LNM == 3) wave_epilogue_split<T16, 1>(p, 0, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, stg, stg_edge, lane_e);
        else wave_epilogue<OUT_F32, T16, EM, 1>(p, 0, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, csum_l, stg, nullptr, m0, stg_edge, sr0, sr1, lane_e);
#ifdef LA_TILE_STAMPS
        if (tid == 0 && g_tile_stamps) {       // per tile: top of the iteration, first fragments in, main loop done, epilogue begins / ends
            unsigned long long *o = g_tile_stamps + (size_t)t * 8;
            unsigned hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
            o[0] = ps0; o[1] = ps1; o[2] = ps2; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = hw_id; o[5] = xcc_id; o[6] = ps3; o[7] = prefetched ? 1 : 0;
        }
#endif
        if (tn < 0) break;
        if (chain) {
            c = cn; m0 = mn0; n0 = nn0; bias_l = bias_n; csum_l = csum_n; sr0 = sn0; sr1 = sn1;
        } else {
            __syncthreads();                                 // an edge tile staged at the ring's front: everyone is done with it
            open_tile(tn, c, m0, n0, bias_l, csum_l, sr0, sr1);
        }
        prefetched = chain;
        t = tn;
    }
    // the last workgroup out re-arms the counters for the next launch on this stream
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&tk->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) __hip_atomic_store(&tk->head[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tk->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Epilogue of the one-wave-per-SIMD kernel: one wave's 128x128 tile, accumulators in the AGPRs.  Per pass of 32 rows the wave
// writes its accumulator tuples STRAIGHT from the AGPRs into its own 32 x 528 bytes of LDS (ds_write_b128 takes AGPR data: no
// v_accvgpr_read, no VGPR copy of the tile) and reads them back row-major -- lane (lane >> 5, lane & 31) = (row parity, four
// consecutive columns), two whole 512-byte rows per instruction -- where the LayerNorm fold / bias, GELU, the f32 residual,
// the stores, the 16-bit copy and the per-segment row statistics are applied in the same order and with the same operations
// as wave_epilogue (bit-identical results).  The lane's four columns are fixed, so its bias / column-sum values (b4, cs4) are
// loaded once by the caller before the main loop; the pass's residual rows and LayerNorm row statistics are requested before
// the staging, 16 rows in flight per lane.
constexpr int MONO_PITCH = 528;
template <int OFF> __device__ __forceinline__ void ds_write128_agpr(unsigned addr, const f32x4 &a) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "a"(a), "n"(OFF) : "memory");
}
template <bool OUT_F32, typename T16, int LNM>
__device__ __forceinline__ void mono_epilogue(const GemmParams &p, int z, f32x4 (&acc)[8][8], int wrow0, int wcol0, bool has_bias,
                                              const float (&b4)[4], const float (&cs4)[4], unsigned char *reg) {
    constexpr int PITCH = MONO_PITCH;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4, c = lane & 31, rp = lane >> 5;
    typedef typename std::conditional<OUT_F32, float, T16>::type TC;
    TC *C = reinterpret_cast<TC *>(p.C) + (int64_t)z * p.strideC;
    const float *R = p.residual ? p.residual + (int64_t)z * p.strideR : nullptr;
    const bool do_gelu = p.epilogue & LA_EPI_GELU;
    const bool do_res = (p.epilogue & LA_EPI_RESIDUAL) && R;
    const bool fast_c = ((p.ldc * (int64_t)sizeof(TC)) % 16 == 0) && ((uintptr_t)C % 16 == 0);
    const bool fast_r = do_res && (p.ldr % 4 == 0) && ((uintptr_t)R % 16 == 0);
    const bool full = wrow0 + 128 <= p.M && wcol0 + 128 <= p.N && fast_c && (!do_res || fast_r);     // wave-uniform
    const int n = wcol0 + c * 4;
    T16 *C2 = nullptr;
    if constexpr (LNM == 1 && OUT_F32) C2 = reinterpret_cast<T16 *>(p.C2) + (int64_t)z * p.strideC;
    const unsigned wr_addr = la::lds_addr_u32(reg) + (unsigned)(r * PITCH + q * 16);
    const unsigned char *rd = reg + rp * PITCH + c * 16;
    // the 16 accumulator tuples of pass h -> LDS (the AGPR names are compile-time: one arm per pass)
    auto stage = [&](auto hc) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
        la::gemm::static_for<0, 16>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, mm = i >> 3, ni = i & 7;
            ds_write128_agpr<mm * 16 * PITCH + ni * 64>(wr_addr, acc[2 * h + mm][ni]);
        });
    };
    // one pass of 32 rows; the row code exists once per (FULL, RES) -- the passes are a run-time loop around it, so the whole
    // epilogue stays a few thousand instructions (fully unrolled it was 45 k: every wave then streams its code from L2)
    auto pass = [&](int h, auto fullc, auto resc) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(fullc)::value, RES = decltype(resc)::value;
        const int row0 = wrow0 + h * 32;
        float4 t[16];
        float2 st[16];
        // (row addresses by pointer increments: a 64-bit multiply per row costs more vector cycles than the row's arithmetic)
        if constexpr (FULL && RES) {
            const float *rr = R + (int64_t)(row0 + rp) * p.ldr + n;
            const int64_t rstep = 2 * p.ldr;
#pragma unroll
            for (int it = 0; it < 16; ++it) { t[it] = *reinterpret_cast<const float4 *>(rr); rr += rstep; }
        }
        const int64_t cstep = 2 * p.ldc;
        int64_t off = (int64_t)(row0 + rp) * p.ldc + n - cstep;
        if constexpr (LNM == 2) {
#pragma unroll
            for (int it = 0; it < 16; ++it) st[it] = reinterpret_cast<const float2 *>(p.ln_stats)[min(row0 + it * 2 + rp, p.M - 1)];
        }
        switch (h) {
            case 0: stage(std::integral_constant<int, 0>{}); break;
            case 1: stage(std::integral_constant<int, 1>{}); break;
            case 2: stage(std::integral_constant<int, 2>{}); break;
            default: stage(std::integral_constant<int, 3>{}); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(rd + it * 2 * PITCH);
            const int m = row0 + it * 2 + rp;
            off += cstep;
            if constexpr (LNM == 2) {
                const float rs = st[it].y, bm = -st[it].x * st[it].y;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], rs, fmaf(bm, cs4[j], b4[j]));
            } else if (has_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += b4[j];
            }
            if (do_gelu) {
                if constexpr (!OUT_F32) {
                    const la::f32x2 lo = la::gelu_sig2(la::f32x2{v[0], v[1]}), hi = la::gelu_sig2(la::f32x2{v[2], v[3]});
                    v = f32x4{lo.x, lo.y, hi.x, hi.y};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = la::gelu_erf(v[j]);
                }
            }
            if constexpr (FULL) {
                if constexpr (RES) { v[0] += t[it].x; v[1] += t[it].y; v[2] += t[it].z; v[3] += t[it].w; }
                if ((p.epilogue & 2048) && v[0] != 12345.678f) continue;      // developer probe: the epilogue without its stores
                if constexpr (sizeof(TC) == 4) {
                    *reinterpret_cast<float4 *>(C + off) = make_float4(v[0], v[1], v[2], v[3]);
                    if constexpr (LNM == 1) {
                        const ushort4 pk = la::Pack4<T16>::run(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<ushort4 *>(C2 + off) = pk;
                        if (p.ln_part) {
                            const float2 sg = segment_stats<T16>(pk);
                            if (r == 0) reinterpret_cast<float2 *>(p.ln_part)[(int64_t)((wcol0 >> 6) + (c >> 4)) * p.M + m] = sg;
                        }
                    }
                } else {
                    *reinterpret_cast<ushort4 *>(C + off) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                }
            } else {
                // ragged tile / unaligned operands (wave_epilogue's generic path, same order of operations)
                const bool in = m < p.M && n < p.N;
                const int nv = in ? min(4, p.N - n) : 0;
                if (do_res && in) {
                    const float *rr = R + (int64_t)m * p.ldr + n;
                    if (fast_r && nv == 4) {
                        const float4 tt = *reinterpret_cast<const float4 *>(rr);
                        v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w;
                    } else {
                        for (int j = 0; j < nv; ++j) v[j] += rr[j];
                    }
                }
                if (in) {
                    TC *cc = C + off;
                    if (fast_c && nv == 4) {
                        if constexpr (sizeof(TC) == 4) *reinterpret_cast<float4 *>(cc) = make_float4(v[0], v[1], v[2], v[3]);
                        else *reinterpret_cast<ushort4 *>(cc) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                    } else {
                        for (int j = 0; j < nv; ++j) la::Elem<TC>::store(cc + j, v[j]);
                    }
                }
                if constexpr (LNM == 1) {
                    if (in) {
                        for (int j = 0; j < nv; ++j) la::Elem<T16>::store(C2 + off + j, v[j]);
                    }
                    if (p.ln_part) {       // N % 64 == 0 on this path (host check): a segment's 16 lanes are in or out together
                        const float2 sg = segment_stats<T16>(la::Pack4<T16>::run(v[0], v[1], v[2], v[3]));
                        if (r == 0 && in) reinterpret_cast<float2 *>(p.ln_part)[(int64_t)((wcol0 >> 6) + (c >> 4)) * p.M + m] = sg;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // one pass's loads and rows in flight at a time (register budget)
    };
    typedef std::true_type TT;
    typedef std::false_type FF;
    if (full) {
        if (do_res) {
#pragma nounroll
            for (int h = 0; h < 4; ++h) pass(h, TT{}, TT{});
        } else {
#pragma nounroll
            for (int h = 0; h < 4; ++h) pass(h, TT{}, FF{});
        }
    } else {
#pragma nounroll
        for (int h = 0; h < 4; ++h) pass(h, FF{}, FF{});
    }
}

// The one-wave-per-SIMD kernel (LA_PP_DBG=73): 256x256 tile, 4 waves x 128x128 wave tiles, hand-placed main loop
// (mainloop_mono_asm), accumulators in the AGPRs from the first MFMA to the epilogue's ds_write.
template <bool OUT_F32, typename T16, int LNM = 0>
__global__ __launch_bounds__(MONO::THREADS, 1) void gemm_mono_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord(tile, p.tiles_m, p.tiles_n, p.group);
    const int m0 = tc.tm * 256, n0 = tc.tn * 256;
    const int z = blockIdx.y;
    const T16 *A = reinterpret_cast<const T16 *>(p.A) + (int64_t)z * p.strideA;
    const T16 *W = reinterpret_cast<const T16 *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    // per-column epilogue operands of this lane's four columns, requested before the main loop
    float b4[4], cs4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ncol = min(n0 + wc * 128 + (lane & 31) * 4 + j, p.N - 1);
        b4[j] = has_bias ? bias[ncol] : 0.f;
        cs4[j] = 0.f;
        if constexpr (LNM == 2) cs4[j] = p.ln_csum[ncol];
    }
    f32x4 acc[8][8];
    mainloop_mono_asm<T16>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc);
    if (p.epilogue & 256) return;            // developer probe (KB_NOSTORE): the main loop (volatile asm: not removable) alone
    mono_epilogue<OUT_F32, T16, LNM>(p, z, acc, m0 + wr * 128, n0 + wc * 128, has_bias, b4, cs4, lds + wave * (32 * MONO_PITCH));
}

template <bool OUT_F32, typename T16, int LNM = 0>
int launch_mono(GemmParams p, int batch, hipStream_t stream) {
    auto kern = gemm_mono_kernel<OUT_F32, T16, LNM>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, MONO::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, 256);
    p.tiles_n = la::cdiv(p.N, 256);
    p.group = getenv("LA_GEMM_GROUP") ? std::min(p.group, p.tiles_n) : std::min(p.tiles_n, std::max(4, p.group / 2));
    la::TimerScope ts("gemm_bf16", stream, 2.0 * p.M * p.N * p.K * batch);
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(MONO::THREADS), MONO::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// the one-wave-per-SIMD kernel by epilogue mode (the same three the ping-pong kernel has)
template <bool OUT_F32, typename T16>
int launch_mono_modes(GemmParams p, int batch, hipStream_t stream) {
    if (p.ln_stats) return launch_mono<OUT_F32, T16, 2>(p, batch, stream);
    if constexpr (OUT_F32) {
        if (p.C2) return launch_mono<OUT_F32, T16, 1>(p, batch, stream);
    }
    return launch_mono<OUT_F32, T16, 0>(p, batch, stream);
}

// LA_GEMM_Q4=1 (read per launch): the four-wave, two-workgroups-per-CU form for the shapes the hand-placed loop takes (bf16).
template <bool OUT_F32, typename T16, int LNM, bool WIDE>
int launch_q4(GemmParams p, int batch, hipStream_t stream) {
    auto kern = gemm_q4_kernel<OUT_F32, T16, LNM, 8, WIDE>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q4::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, WIDE ? Q4::TN : Q4::TM);
    p.tiles_n = la::cdiv(p.N, WIDE ? Q4::TM : Q4::TN);
    // column groups of the same WIDTH and row blocks of the same HEIGHT as the 8-wave kernel's (which counts 256 x 256 tiles)
    const int gw = WIDE ? 1 : 2, mh = WIDE ? 2 : 1;
    p.group = getenv("LA_GEMM_GROUP") ? std::min(gw * p.group, p.tiles_n) : std::min(p.tiles_n, gw * std::max(4, p.group / 2));
    p.mblock = p.tiles_n > p.group ? 32 * mh : 0;
    if (const char *g = getenv("LA_GEMM_MBLOCK")) p.mblock = atoi(g);
    if (const char *g = getenv("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    if (const char *g = getenv("LA_GEMM_Q4_PRIO")) { if (atoi(g) == 1) p.epilogue |= LA_EPI_Q4_PRIO; }
    la::TimerScope ts("gemm_bf16", stream, 2.0 * p.M * p.N * p.K * batch);
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(Q4::THREADS), Q4::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// Ticket words of the persistent kernel: one block per (device, stream), zeroed when created (on that stream) and left zeroed by
// every launch's last workgroup.  Launches on one stream are ordered, so a block is never shared by two running kernels.
PersistTickets *persist_tickets(hipStream_t stream) {
    struct Slot { int dev; hipStream_t stream; PersistTickets *ptr; };
    static std::mutex mu;
    static std::vector<Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (const Slot &sl : slots)
        if (sl.dev == dev && sl.stream == stream) return sl.ptr;
    PersistTickets *ptr = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&ptr), sizeof(PersistTickets)) != hipSuccess) return nullptr;
    if (hipMemsetAsync(ptr, 0, sizeof(PersistTickets), stream) != hipSuccess) { (void)hipFree(ptr); return nullptr; }
    slots.push_back(Slot{dev, stream, ptr});
    return ptr;
}

int device_cu_count() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    dev &= 63;
    if (n[dev] == 0) {
        int v = 0;
        n[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n[dev];
}

// The persistent form takes a launch when its tiles outnumber the CUs (there is a next tile to prefetch), the hand-placed main
// loop fits, and every interior wave tile will take the wide epilogue path (row pitches / pointers aligned: checked HERE, because
// the persistent kernel prefetches into the LDS the element-wise path would stage through).  LA_GEMM_PERSIST=0 (read per launch)
// keeps one workgroup per tile: the A/B partner.
template <bool OUT_F32, typename T16, int LNM>
bool persist_eligible(const GemmParams &p, int batch, bool duo) {
    // MEASURED SLOWER than one workgroup per tile (profiles/r4_kbench_persistent_ab.txt, r4_persistent_kernel_timeline.txt; DESIGN.md
    // "GEMM, round 4"): opt-in with LA_GEMM_PERSIST=1 (read per launch), bf16 only.
    const char *e = getenv("LA_GEMM_PERSIST");
    if (!(e && e[0] == '1')) return false;
    if (!std::is_same<T16, bf16_t>::value) return false;
    if (!duo || batch != 1) return false;
    const int nt = la::cdiv(p.M, PP::TM) * la::cdiv(p.N, PP::TN);
    if (nt <= device_cu_count()) return false;
    if (p.N % PP::TN != 0) return false;                                   // (column edges would need the element-wise path too)
    if constexpr (LNM == 3) {
        if (!(p.ldc % 4 == 0 && (uintptr_t)p.C % 4 == 0 && (uintptr_t)p.C2 % 8 == 0)) return false;
        if ((p.epilogue & LA_EPI_RESIDUAL) && p.residual && !(p.ldr % 4 == 0 && (uintptr_t)p.residual % 16 == 0)) return false;
    } else {
        const int64_t es = OUT_F32 ? 4 : 2;
        if (!((p.ldc * es) % 16 == 0 && (uintptr_t)p.C % 16 == 0)) return false;
        if ((p.epilogue & LA_EPI_RESIDUAL) && p.residual && !(p.ldr % 4 == 0 && (uintptr_t)p.residual % 16 == 0)) return false;
    }
    return true;
}

template <bool OUT_F32, typename T16, int LNM>
int launch_pp_persist(GemmParams p, hipStream_t stream) {
    auto kern = gemm_pp_persist_kernel<OUT_F32, T16, LNM>;
    constexpr int LDS_BYTES = 160 * 1024;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_once.mark();
    }
    PersistTickets *tk = persist_tickets(stream);
    if (!tk) { la::set_error("gemm: ticket block allocation failed"); return LA_EHIP; }
    p.tiles_m = la::cdiv(p.M, PP::TM);
    p.tiles_n = la::cdiv(p.N, PP::TN);
    p.group = getenv("LA_GEMM_GROUP") ? std::min(p.group, p.tiles_n) : std::min(p.tiles_n, std::max(4, p.group / 2));
    p.mblock = p.tiles_n > p.group ? 32 : 0;
    if (const char *g = getenv("LA_GEMM_MBLOCK")) p.mblock = atoi(g);
    if (const char *g = getenv("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    const int grid = std::min(p.tiles_m * p.tiles_n, device_cu_count());
    la::TimerScope ts("gemm_bf16", stream, 2.0 * p.M * p.N * p.K);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PP::THREADS), LDS_BYTES, stream, p, tk);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

bool lab_try_launch(GemmParams &p, int batch, bool out_f32, int lnm, bool duo, hipStream_t stream, int *rc) {
    typedef bf16_t T;
    const char *dbg_env = getenv("LA_PP_DBG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    const bool fits = p.K % 128 == 0 && p.K >= 256;
    if (dbg == 73 && fits && lnm != 3 && lnm != 4) {                     // one wave per SIMD
        *rc = out_f32 ? launch_mono_modes<true, T>(p, batch, stream) : launch_mono_modes<false, T>(p, batch, stream);
        return true;
    }
    if (lnm == 4) {                                                      // in-loop row statistics
        if (!duo) {
            la::set_error("gemm_fused_ln: in-loop row statistics need the hand-placed main loop (K %% 128 == 0, K >= 256; K = %d)", p.K);
            *rc = LA_EUNSUPPORTED;
            return true;
        }
        *rc = out_f32 ? launch_pp_loop<true, true, T, 4>(p, batch, stream) : launch_pp_loop<false, true, T, 4>(p, batch, stream);
        return true;
    }
    if (const char *q4 = getenv("LA_GEMM_Q4"); q4 && duo && Q4::rem_of(p.K / 32) == 8 && ((lnm == 3 && out_f32) || ((lnm == 0 || lnm == 2) && !out_f32))) {
        const int f = atoi(q4);                                          // (K = 256, 1024, 4096, ..); 1 = 256 x 128 tiles, 2 = 128 x 256
        if (f == 1 || f == 2) {
            if (lnm == 3) *rc = f == 1 ? launch_q4<true, T, 3, false>(p, batch, stream) : launch_q4<true, T, 3, true>(p, batch, stream);
            else if (lnm == 2) *rc = f == 1 ? launch_q4<false, T, 2, false>(p, batch, stream) : launch_q4<false, T, 2, true>(p, batch, stream);
            else *rc = f == 1 ? launch_q4<false, T, 0, false>(p, batch, stream) : launch_q4<false, T, 0, true>(p, batch, stream);
            return true;
        }
    }
    if (lnm == 3 && persist_eligible<true, T, 3>(p, batch, duo)) { *rc = launch_pp_persist<true, T, 3>(p, stream); return true; }
    if (lnm == 2 && !out_f32 && persist_eligible<false, T, 2>(p, batch, duo)) { *rc = launch_pp_persist<false, T, 2>(p, stream); return true; }
    if (lnm == 0 && !out_f32 && persist_eligible<false, T, 0>(p, batch, duo)) { *rc = launch_pp_persist<false, T, 0>(p, stream); return true; }
    return false;
}

int lab_set_tile_stamps(void *buf) {
#ifdef LA_TILE_STAMPS
    return set_tile_stamps_here(buf);
#else
    (void)buf;
    return LA_OK;
#endif
}

}  // namespace gemm
}  // namespace la
