// la_gemm_pp_kernel.h -- the 256 x 256 GEMM kernel of the 16-bit modes (every large Linear / conv-as-GEMM of the encoder, the GRU
// input projections) and its launchers, as templates over the operand type: instantiated by la_gemm_pp_bf16.hip and
// la_gemm_pp_f16.hip (two translation units, so the two operand types compile in parallel).
// Main loops: la_gemm_pp.h (hand-placed flat stream / quadrant ping-pong).  Epilogues: la_gemm_epilogue.h.
#pragma once
#include <algorithm>

#include "la_gemm_epilogue.h"

namespace la {
namespace gemm {

#ifdef LA_TILE_STAMPS
// Diagnostic build only (tools/tile_timeline.py): per workgroup (wall clock at entry, after the prologue, after the main loop, at
// the end; HW_ID) into a buffer that nothing else reads -- where a tile's lifetime goes and how long a CU waits for its next one.
static __device__ unsigned long long *g_tile_stamps = nullptr;
static inline int set_tile_stamps_here(void *buf) {
    unsigned long long *b = static_cast<unsigned long long *>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), &b, sizeof(b)) == hipSuccess ? LA_OK : LA_EHIP;
}
#endif

template <bool OUT_F32, bool DUO, typename T16, int LNM = 0>
__global__ __launch_bounds__(PP::THREADS, 2) void gemm_pp_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_t1 = stamp_t0;
#endif
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord_mb(tile, p.tiles_m, p.tiles_n, p.group, p.mblock);
    const int m0 = tc.tm * PP::TM, n0 = tc.tn * PP::TN;
    const int z = blockIdx.y;
    const T16 *A = reinterpret_cast<const T16 *>(p.A) + (int64_t)z * p.strideA;
    const T16 *W = reinterpret_cast<const T16 *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    // Per-column epilogue operands are requested BEFORE the main loop, one column per lane (this wave's 64 columns), and
    // handed to the row-major epilogue layout with ds_bpermute afterwards: all eight waves reach the epilogue together, so a load
    // issued there is a fully exposed L2 round trip per tile.
    const int ncol = min(n0 + wc * 64 + lane, p.N - 1);
    const float bias_l = has_bias ? bias[ncol] : 0.f;
    float csum_l = 0.f;
    if constexpr (LNM == 2 || LNM == 4 || LNM == 6) csum_l = p.ln_csum[ncol];
    // LNM == 2: the tile's 256 row statistics are requested here too (32 rows per wave) and handed to the epilogue through LDS after
    // the main loop: loaded inside the epilogue they were one exposed L2 round trip per pass of 32 rows, four per tile.
    float2 st_pre = make_float2(0.f, 1.f);
    if constexpr (LNM == 2) {
        if (lane < 32) st_pre = reinterpret_cast<const float2 *>(p.ln_stats)[min(m0 + wr * 128 + wc * 32 + lane, p.M - 1)];
    }
    if constexpr (LNM == 6) {                                                 // f16x2 products: (0, row scale) in the statistics' place
        if (lane < 32) st_pre = make_float2(0.f, p.ln_stats[min(m0 + wr * 128 + wc * 32 + lane, p.M - 1)]);
    }

    f32x4 acc[8][4];
#ifdef LA_TILE_STAMPS
#define LA_STAMP_ARG , stamp_t1
#else
#define LA_STAMP_ARG
#endif
    float sacc[4] = {0.f, 0.f, 0.f, 0.f};
    float2 *stats_tab = reinterpret_cast<float2 *>(lds + 120 * 1024);        // LNM == 4: inside ring slot 3, clear of the epilogue staging
    if constexpr (LNM == 4) {
        static_assert(DUO || LNM != 4, "the main loop takes the row statistics only in its hand-placed form");
        const int wc_u = __builtin_amdgcn_readfirstlane(wc);
        switch (wc_u) {                                                       // (the fragment registers are named at compile time)
            case 0: mainloop_duo_asm<T16, 0>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
            case 1: mainloop_duo_asm<T16, 1>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
            case 2: mainloop_duo_asm<T16, 2>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
            default: mainloop_duo_asm<T16, 3>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG, sacc); break;
        }
        // sum over the four lanes that hold one row's four k chunks, then (mean, rstd) of rows wr * 128 + (2 wc + i) * 16 + r
        const float inv_k = 1.0f / (float)p.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s1 = sacc[2 * i], s2 = sacc[2 * i + 1];
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            const float mean = s1 * inv_k;
            const float var = fmaxf(fmaf(-mean, mean, s2 * inv_k), 0.f);
            if (lane < 16) stats_tab[wr * 128 + (2 * wc + i) * 16 + lane] = make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
        }
    } else if constexpr (LNM == 6) {
        static_assert(DUO || LNM != 6, "the segmented-K loop is the hand-placed one");
        mainloop_duo_seg_asm<T16>(A, p.lda, p.M, W, p.ldw, p.N, p.K, p.plane_a, p.plane_w, m0, n0, lds, acc);
    } else if constexpr (DUO) {
        mainloop_duo_asm<T16>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc LA_STAMP_ARG);
    } else {
        mainloop_pp<T16>(A, p.lda, p.M, W, p.ldw, p.N, p.K, m0, n0, lds, acc);
    }
    if constexpr (LNM == 2 || LNM == 6) {
        if (lane < 32) stats_tab[wr * 128 + wc * 32 + lane] = st_pre;       // (both main loops end behind a barrier: slot 3 is free)
    }
#ifdef LA_TILE_STAMPS
    const unsigned long long stamp_t2 = __builtin_amdgcn_s_memrealtime();
#endif

    __syncthreads();
    if constexpr (LNM == 3) {
        // 16-row passes with the residual rows requested one pass ahead (default), or the 32-row passes (LA_EPI_SPLIT_PASS=32: A/B)
        if (p.epilogue & LA_EPI_SPLIT_PASS32) wave_epilogue_split<T16, 0>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, lds + wave * (32 * EPI_PITCH));
        else wave_epilogue_split<T16, 1>(p, z, acc, m0 + wr * 128, n0 + wc * 64, has_bias, bias_l, lds + wave * (32 * EPI_PITCH), lds + wave * (32 * EPI_PITCH));
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

template <bool OUT_F32, bool DUO, typename T16, int LNM = 0>
int launch_pp_loop(GemmParams p, int batch, hipStream_t stream) {
    auto kern = gemm_pp_kernel<OUT_F32, DUO, T16, LNM>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, PP::TM);
    p.tiles_n = la::cdiv(p.N, PP::TN);
    // column tiles that walk the M dimension together (their W panels share the XCD's L2 with the streaming A panel).
    // At least 4: with fewer, the K=4096 GEMM (N = 4 tiles) re-reads its 2 MB-per-row-block A panel once per column tile
    // (in-pipeline sweep: 1 -> 46.1 ms/step, 4 -> 45.7, 8 -> 45.9, 16 -> 46.4).
    p.group = la::dev_env("LA_GEMM_GROUP") ? std::min(p.group, p.tiles_n) : std::min(p.tiles_n, std::max(4, p.group / 2));
    // More than one column group: the groups follow each other inside blocks of 32 row tiles (tile_coord_mb), so a block's A panels
    // (16 MB at K = 1024) are re-read per group out of the Infinity Cache instead of once per sweep over all of M.  Alone on the chip
    // with cold A the MLP-up shape runs 438 -> 352 us (tools/kbench.py order); inside the pipeline, where A was just written, 0.6 %
    // of the step (profiles/r4_ab_mblock.txt).
    // (the f16x2 form -- LNM 6 -- stages both planes of every operand row: twice the bytes per row panel, and blocks of 8 row tiles measured best there:
    //  QKV 767 -> 749 us, MLP-up 1036 -> 1015 us at 48000 rows, the N = 1024 shapes flat; profiles/r6_x2_tile_order.txt)
    p.mblock = p.tiles_n > p.group ? (LNM == 6 ? 8 : 32) : 0;
    // (experiment build only: developer sweeps / probes, read per launch)
    if (const char *g = la::dev_env("LA_GEMM_MBLOCK")) p.mblock = atoi(g);
    if (const char *g = la::dev_env("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    if (const char *g = la::dev_env("LA_EPI_PROBE")) p.epilogue |= (atoi(g) & 7) << 16;
    // (LNM 6: the f16x2 form of a float32 product -- its own timer family, counted at its ALGORITHMIC flops: a third of the MFMA work)
    la::TimerScope ts(LNM == 6 ? "gemm_f16x2" : "gemm_bf16", stream, 2.0 * p.M * p.N * p.K * batch);
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(PP::THREADS), PP::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// Main loop: the hand-placed flat stream (mainloop_duo_asm) where its k-step structure fits -- K a multiple of 128 (four k-steps
// of 32 per ring turn), at least 256 -- else the quadrant ping-pong; option gemm_loop = 99 (LA_PP_DBG) forces the ping-pong
// everywhere.  Same tile, same accumulation order, same epilogue arithmetic: identical bits.
template <bool OUT_F32, typename T16>
int launch_pp(GemmParams p, int batch, hipStream_t stream) {
    const bool duo = p.K % 128 == 0 && p.K >= 256 && la::opts().gemm_loop != 99;
    const int lnm = p.ln_stats ? 2 : (p.ln_csum ? 4 : ((OUT_F32 && p.C2) ? 1 : 0));
#ifdef LA_EXPERIMENTS
    if constexpr (std::is_same<T16, bf16_t>::value) {
        int rc = LA_OK;
        if (lab_try_launch(p, batch, OUT_F32, lnm, duo, stream, &rc)) return rc;
    }
#endif
    if (lnm == 4) {
        la::set_error("gemm_fused_ln: the in-loop row statistics (ln_csum without ln_stats) were measured slower than la_row_stats16 and live "
                      "in the experiment build only (tools/build_variant.sh, bfloat16)");
        return LA_EUNSUPPORTED;
    }
    if (lnm == 2) return duo ? launch_pp_loop<OUT_F32, true, T16, 2>(p, batch, stream) : launch_pp_loop<OUT_F32, false, T16, 2>(p, batch, stream);
    if constexpr (OUT_F32) {
        if (lnm == 1) return duo ? launch_pp_loop<OUT_F32, true, T16, 1>(p, batch, stream) : launch_pp_loop<OUT_F32, false, T16, 1>(p, batch, stream);
    }
    return duo ? launch_pp_loop<OUT_F32, true, T16>(p, batch, stream) : launch_pp_loop<OUT_F32, false, T16>(p, batch, stream);
}

// la_gemm_split's launch: LNM = 3, (hi, lo) out
template <typename T16>
int launch_split(GemmParams p, int batch, hipStream_t stream) {
    const bool duo = p.K % 128 == 0 && p.K >= 256 && la::opts().gemm_loop != 99;
    if (const char *e = la::dev_env("LA_EPI_SPLIT_PASS")) { if (atoi(e) == 32) p.epilogue |= LA_EPI_SPLIT_PASS32; }
#ifdef LA_EXPERIMENTS
    if constexpr (std::is_same<T16, bf16_t>::value) {
        int rc = LA_OK;
        if (lab_try_launch(p, batch, true, 3, duo, stream, &rc)) return rc;
    }
#endif
    return duo ? launch_pp_loop<true, true, T16, 3>(p, batch, stream) : launch_pp_loop<true, false, T16, 3>(p, batch, stream);
}

}  // namespace gemm
}  // namespace la
