// la_gemm_params.h -- launch parameters of the GEMM kernels and the entry points of the translation units they are built in:
// la_gemm.hip (host dispatch, the 128 x 128 kernel family incl. float32, split-K), la_gemm_pp_bf16.hip / la_gemm_pp_f16.hip (the
// 256 x 256 kernel per operand type: compiled in parallel), lab/la_gemm_lab.hip (experiment build only).
#pragma once
#include "la_common.h"

namespace la {
namespace gemm {

struct GemmParams {
    int M, N, K;
    const void *A;
    int64_t lda, strideA;
    const void *W;
    int64_t ldw, strideW;
    void *C;
    int64_t ldc, strideC;
    const float *bias;
    int64_t strideBias;
    const float *residual;
    int64_t ldr, strideR;
    int epilogue;
    int tiles_m, tiles_n, group;
    int mblock = 0;                    // ping-pong kernel: row tiles per M block of the tile order (tile_coord_mb; 0 = column groups over all of M)
    // LayerNorm folded into the GEMMs around it (ping-pong kernel only; la_gemm_fused_ln):
    void *C2 = nullptr;                // producer: second, 16-bit copy of the f32 result rows (the next GEMM's raw A operand)
    int64_t ldc2 = 0, strideC2 = 0;
    const float *ln_stats = nullptr;   // consumer: per-row (mean, rstd) of the raw A rows, [M][2]
    const float *ln_csum = nullptr;    // consumer: c[n] = sum_k W'[n][k] of the gamma-folded weights, [N]
    int K_tail = 0;                    // split-K: K of the LAST batch slot when the chunks are uneven (0 = p.K)
    int64_t plane_a = 0, plane_w = 0;  // f16x2 products (LNM 6): pitch in elements between the hi and the lo plane inside a row of A / W;
                                       // there ln_stats = per-row scales [M] and ln_csum = per-column scales [N] (powers of two)
    float *ln_part = nullptr;          // producer (optional): per-row partial statistics of the 16-bit copy, [N/64][M][2] =
                                       // (mean, sum of squared deviations) of each 64-column segment (la_ln_stats_finalize)
};

// internal epilogue bits (beside the public LA_EPI_* of lyricalign.h)
constexpr int LA_EPI_SPLIT_INPLACE = 1 << 20;    // the residual is the split stream itself
constexpr int LA_EPI_SPLIT_PASS32 = 1 << 21;     // (experiment build, LA_EPI_SPLIT_PASS=32) the 32-row passes without the one-pass-ahead requests
constexpr int LA_EPI_Q4_PRIO = 1 << 23;          // (experiment build, LA_GEMM_Q4_PRIO=1) gemm_q4_kernel's prologue / epilogue at wave priority 3

// The 256 x 256 kernel by operand type (dtype: LA_BF16 | LA_F16).  launch_pp16: plain / LayerNorm-producer (p.C2) / LayerNorm-consumer
// (p.ln_stats) epilogues by p's fields; launch_split16: the split residual stream's producers (la_gemm_split; p.C = lo, p.C2 = hi).
int launch_pp_bf16(GemmParams p, int batch, bool out_f32, hipStream_t stream);
int launch_pp_f16(GemmParams p, int batch, bool out_f32, hipStream_t stream);
int launch_split_bf16(GemmParams p, int batch, hipStream_t stream);
int launch_split_f16(GemmParams p, int batch, hipStream_t stream);
// la_gemm_f16x2 (la_f32x2.hip): float32 products as three f16 products over segmented K, f32 out, scale epilogue; batch = split-K slots
int launch_x2_f16(GemmParams p, int batch, hipStream_t stream);

#ifdef LA_EXPERIMENTS
// lab/la_gemm_lab.hip: the measured-slower structures behind their per-launch developer switches (LA_GEMM_Q4, LA_GEMM_PERSIST,
// LA_PP_DBG=73, the in-loop LayerNorm statistics).  true = the launch was taken over (*rc = its status).
bool lab_try_launch(GemmParams &p, int batch, bool out_f32, int lnm, bool duo, hipStream_t stream, int *rc);
int lab_set_tile_stamps(void *buf);
#endif

}  // namespace gemm
}  // namespace la
