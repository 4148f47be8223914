// la_head.hip -- fused head tail: Linear(2H -> V) + emission prep without ever writing the
// [batch][frames][V] logits (126.8 MB per 30 s clip in the reference, which then copies them to
// the host: module/align_model.py:38, inference_alignment.py:161, utils/alignment.py:123-134).
//
//  1. gather_rows:   per clip, the <= Lmax+1 rows of W_fc / b_fc the DP will read (silence column +
//                    one column per label) -> Wg [B][Lmax+1][K]
//  2. batched GEMM:  raw[b][t][s] = act[b][t][:] . Wg[b][s][:] + bg[b][s]        (la_gemm core)
//  3. lse GEMM:      the full [B*T] x V product, but the epilogue only reduces each 64-column
//                    accumulator strip to (max, sum exp(x - max)) per row        (this file)
//  4. merge:         per row, combine the 2*ceil(V/128) partials into the log-normaliser and apply
//                    the reference's emission formulas to raw                     (this file)
// Algorithmic traffic: reads act + W_fc once, writes 8 B per row per 64 columns; 2*M*V*K flops.
#include "la_gemm_core.h"
#include "la_gemm_pp.h"

using la::bf16_t;
using namespace la::gemm;

namespace {

constexpr float kLog2e = 1.4426950408889634f;

struct LseParams {
    int M, N, K;
    const void *A;
    int64_t lda;
    const void *W;
    const float *bias;
    float2 *partials;  // [M][nparts]: one (max, sum exp) pair per row and 64-column strip
    int nparts;
    int lo, hi;        // inclusive column range of the normaliser
    int tiles_m, tiles_n, group;
    // float32 rows on the f16x2 products (fc_lse_x2_kernel): A / W are half planes [.][2][K], sa / sw their per-row inverse scales
    const float *sa = nullptr, *sw = nullptr;
};

template <typename T, typename C>
__global__ __launch_bounds__(C::THREADS, 2) void fc_lse_kernel(LseParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const TileCoord tc = tile_coord(tile, p.tiles_m, p.tiles_n, p.group);
    const int tn = tc.tn;
    const int m0 = tc.tm * C::TM, n0 = tn * BN;
    f32x4 acc[4][4];
    mainloop<T, C>(reinterpret_cast<const T *>(p.A), p.lda, p.M, reinterpret_cast<const T *>(p.W), p.K, p.N, p.K, m0, n0, lds, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + ni * 16 + q * 4 + j;
                const bool ok = n >= p.lo && n <= p.hi;
                const float x = ok ? acc[mi][ni][j] + p.bias[n < p.N ? n : p.N - 1] : -INFINITY;
                v[ni * 4 + j] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        if (mx > -INFINITY) {
            const float mb = mx * kLog2e;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(fmaf(v[i], kLog2e, -mb));  // exp2(-inf) = 0 for masked columns
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const int m = m0 + wm * 64 + mi * 16 + r;
        if (q == 0 && m < p.M) p.partials[(int64_t)m * p.nparts + tn * 2 + wn] = make_float2(mx, sum);
    }
}

// The same reduction on the 256x256 ping-pong main loop (bf16, large row counts): a wave owns 128 rows x one 64-column strip.
// DUO: the hand-placed flat main loop (mainloop_duo_asm; K a multiple of 128, >= 256) instead of the quadrant ping-pong -- same
// tile, same accumulation order, same bits.
template <typename T16, bool DUO = false>
__global__ __launch_bounds__(PP::THREADS, 2) void fc_lse_pp_kernel(LseParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const TileCoord tc = tile_coord(tile, p.tiles_m, p.tiles_n, p.group);
    const int tn = tc.tn;
    const int m0 = tc.tm * PP::TM, n0 = tn * PP::TN;
    f32x4 acc[8][4];
#ifdef LA_TILE_STAMPS
    unsigned long long stamp_unused = 0;
    if constexpr (DUO) mainloop_duo_asm<T16>(reinterpret_cast<const T16 *>(p.A), p.lda, p.M, reinterpret_cast<const T16 *>(p.W), p.K, p.N, p.K, m0, n0, lds, acc, stamp_unused);
#else
    if constexpr (DUO) mainloop_duo_asm<T16>(reinterpret_cast<const T16 *>(p.A), p.lda, p.M, reinterpret_cast<const T16 *>(p.W), p.K, p.N, p.K, m0, n0, lds, acc);
#endif
    else mainloop_pp<T16>(reinterpret_cast<const T16 *>(p.A), p.lda, p.M, reinterpret_cast<const T16 *>(p.W), p.K, p.N, p.K, m0, n0, lds, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    float b4[4][4];
    bool ok4[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + ni * 16 + q * 4 + j;
            ok4[ni][j] = n >= p.lo && n <= p.hi;
            b4[ni][j] = p.bias[n < p.N ? n : p.N - 1];
        }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = ok4[ni][j] ? acc[mi][ni][j] + b4[ni][j] : -INFINITY;
                v[ni * 4 + j] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        if (mx > -INFINITY) {
            const float mb = mx * kLog2e;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(fmaf(v[i], kLog2e, -mb));
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const int m = m0 + wr * 128 + mi * 16 + r;
        if (q == 0 && m < p.M) p.partials[(int64_t)m * p.nparts + tn * 4 + wc] = make_float2(mx, sum);
    }
}

// The same reduction for FLOAT32 rows on the f16 matrix pipe at float32 accuracy (la_f32x2.hip): act and W_fc as half planes [.][2][K] with
// per-row power-of-two scales, three f16 products per k-step in the segmented main loop (mainloop_duo_seg_asm), the scales applied to the
// accumulators before the bias.  K a multiple of 128, >= 256.
__global__ __launch_bounds__(PP::THREADS, 2) void fc_lse_x2_kernel(LseParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const TileCoord tc = tile_coord(tile, p.tiles_m, p.tiles_n, p.group);
    const int tn = tc.tn;
    const int m0 = tc.tm * PP::TM, n0 = tn * PP::TN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    f32x4 acc[8][4];
    mainloop_duo_seg_asm<la::f16_t>(reinterpret_cast<const la::f16_t *>(p.A), 2 * (int64_t)p.K, p.M, reinterpret_cast<const la::f16_t *>(p.W),
                                    2 * (int64_t)p.K, p.N, p.K, p.K, p.K, m0, n0, lds, acc);
    // (per-column / per-row operands are loaded AFTER the main loop: 56 more live registers across it spill the hand-placed loop's fragments)
    float b4[4][4], sw4[4][4];
    bool ok4[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + ni * 16 + q * 4 + j;
            const int nc = n < p.N ? n : p.N - 1;
            ok4[ni][j] = n >= p.lo && n <= p.hi;
            b4[ni][j] = p.bias[nc];
            sw4[ni][j] = p.sw[nc];
        }
    float sa8[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) sa8[mi] = p.sa[min(m0 + wr * 128 + mi * 16 + r, p.M - 1)];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = ok4[ni][j] ? acc[mi][ni][j] * (sa8[mi] * sw4[ni][j]) + b4[ni][j] : -INFINITY;
                v[ni * 4 + j] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        if (mx > -INFINITY) {
            const float mb = mx * kLog2e;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(fmaf(v[i], kLog2e, -mb));
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const int m = m0 + wr * 128 + mi * 16 + r;
        if (q == 0 && m < p.M) p.partials[(int64_t)m * p.nparts + tn * 4 + wc] = make_float2(mx, sum);
    }
}

// Wg[b][s][:] = W_fc[col(b, s)][:], bg[b][s] = b_fc[col(b, s)];  col = silence column for s = 0, labels[b][s-1] otherwise
template <typename T>
__global__ void gather_rows_kernel(const T *w, const float *bias, int K, int vocab, int variant, const int32_t *labels,
                                   int labels_stride, const int32_t *n_labels, int max_labels, T *wg, float *bg) {
    const int s = blockIdx.x, b = blockIdx.y;
    int col;
    if (s == 0) {
        col = variant == LA_VARIANT_CTC ? vocab - 1 : 0;
    } else {
        const int n = s - 1;
        col = n < n_labels[b] ? labels[(int64_t)b * labels_stride + n] : 0;
        const int hi = variant == LA_VARIANT_CTC ? vocab - 2 : vocab - 1;
        if (col < 1 || col > hi) col = 0;  // out-of-range class id: merge writes -1000 for it
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(w + (int64_t)col * K);
    uint4 *dst = reinterpret_cast<uint4 *>(wg + ((int64_t)b * (max_labels + 1) + s) * K);
    const int nvec = K * (int)sizeof(T) / 16;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) bg[(int64_t)b * (max_labels + 1) + s] = bias[col];
}

// one wave per row: partials -> log-normaliser; raw logits -> emissions (utils/alignment.py:123-134 / :14-20)
__global__ __launch_bounds__(256) void merge_emissions_kernel(const float2 *partials, int nparts, const float *raw,
                                                              int frames, int rows, int variant, int vocab,
                                                              const int32_t *labels, int labels_stride,
                                                              const int32_t *n_labels, int max_labels, float *em,
                                                              int64_t em_bs, int64_t em_rs) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float2 *pr = partials + (int64_t)row * nparts;
    float mx = -INFINITY;
    for (int i = lane; i < nparts; i += 64) mx = fmaxf(mx, pr[i].x);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int i = lane; i < nparts; i += 64) {
        const float2 v = pr[i];
        if (v.y > 0.f) sum += v.y * __builtin_amdgcn_exp2f((v.x - mx) * kLog2e);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float logsum = logf(sum);

    const int b = row / frames, t = row % frames;
    const int L = min(n_labels[b], max_labels);
    const float *x = raw + (int64_t)row * (max_labels + 1);
    float *e = em + (int64_t)b * em_bs + (int64_t)t * em_rs;
    const int32_t *lab = labels + (int64_t)b * labels_stride;
    const int hi = variant == LA_VARIANT_CTC ? vocab - 2 : vocab - 1;
    if (variant == LA_VARIANT_CTC) {
        const float sil = 1.0f / (1.0f + expf(-x[0]));
        const float log_sil = logf(sil), log_voiced = logf(1.0f - sil);  // naive forms on purpose (:125-129)
        if (lane == 0) e[0] = fmaxf(log_sil, -1000.0f);
        for (int n = lane; n < L; n += 64) {
            const int c = lab[n];
            e[1 + n] = (c >= 1 && c <= hi) ? fmaxf(((x[1 + n] - mx) - logsum) + log_voiced, -1000.0f) : -1000.0f;
        }
    } else {
        if (lane == 0) e[0] = fmaxf((x[0] - mx) - logsum, -1000.0f);
        for (int n = lane; n < L; n += 64) {
            const int c = lab[n];
            e[1 + n] = (c >= 1 && c <= hi) ? fmaxf((x[1 + n] - mx) - logsum, -1000.0f) : -1000.0f;
        }
    }
}

struct HeadPlan {
    size_t off_wg, off_bg, off_raw, off_part, off_planes = 0, off_inv = 0, total;
    int tiles_n, nparts;
};

HeadPlan plan_head(int batch, int frames, int in_dim, int vocab, int max_labels, int es, bool x2 = false) {
    HeadPlan pl;
    const int64_t rows = (int64_t)batch * frames, S = max_labels + 1;
    pl.tiles_n = la::cdiv(vocab, BN);
    pl.nparts = 4 * la::cdiv(vocab, PP::TN);     // 64-column strips, rounded up to whole 256-column tiles (>= 2 * tiles_n)
    size_t o = 0;
    pl.off_wg = o;   o += la::round_up((int64_t)batch * S * in_dim * es, 256);
    pl.off_bg = o;   o += la::round_up((int64_t)batch * S * 4, 256);
    pl.off_raw = o;  o += la::round_up(rows * S * 4, 256);
    pl.off_part = o; o += la::round_up(rows * pl.nparts * 8, 256);
    if (x2) {                                   // the half planes of act and their per-row inverse scales
        pl.off_planes = o; o += la::round_up(rows * 2 * in_dim * 2, 256);
        pl.off_inv = o;    o += la::round_up(rows * 4, 256);
    }
    pl.total = o;
    return pl;
}

// Is the float32 normaliser product in the f16x2 kernel's domain?
bool fc_x2_ok(int rows, int in_dim, int vocab) {
    return in_dim % 128 == 0 && in_dim >= 256 && (int64_t)la::cdiv(rows, PP::TM) * la::cdiv(vocab, PP::TN) >= 192;
}

}  // namespace

extern "C" int la_fc_emissions_workspace_bytes(int32_t dtype, int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab,
                                               int32_t max_labels, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && frames > 0 && in_dim > 0 && vocab > 0 && max_labels > 0, "fc_emissions_workspace_bytes: bad arguments");
    *bytes = plan_head(batch, frames, in_dim, vocab, max_labels, dtype == LA_F32 ? 4 : 2).total;
    return LA_OK;
}

extern "C" int la_split_f16x2(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale, void *stream);

// w_x2 / w_x2s (float32 only, or NULL): W_fc as f16x2 planes [V][2][in_dim] + per-row inverse scales -- the normaliser product then runs on
// the f16 matrix pipe at float32 accuracy (fc_lse_x2_kernel) where its shape allows
static int fc_emissions_impl(int32_t dtype, const void *act, int64_t ld_act, const void *w_fc, const float *b_fc, const void *w_x2, const float *w_x2s,
                             int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab, int32_t variant,
                             const int32_t *labels, int32_t labels_stride, const int32_t *n_labels, int32_t max_labels,
                             float *em, int64_t em_batch_stride, int64_t em_row_stride, void *workspace,
                             size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || frames == 0) return LA_OK;
    LA_CHECK_ARG(act && w_fc && b_fc && labels && n_labels && em && workspace, "fc_emissions: null pointer");
    LA_CHECK_ARG(dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16, "fc_emissions: bad dtype");
    LA_CHECK_ARG(batch > 0 && frames > 0 && max_labels > 0, "fc_emissions: bad sizes");
    LA_CHECK_ARG(variant == LA_VARIANT_CTC ? vocab >= 3 : vocab >= 2, "fc_emissions: vocab too small");
    LA_CHECK_ARG(em_row_stride >= max_labels + 1 && labels_stride >= max_labels, "fc_emissions: strides smaller than max_labels");
    const int es = dtype == LA_F32 ? 4 : 2, ke = dtype == LA_F32 ? 32 : 64;
    LA_CHECK_ARG(in_dim % ke == 0, "fc_emissions: in_dim=%d must be a multiple of %d", in_dim, ke);
    LA_CHECK_ARG((ld_act * es) % 16 == 0 && (uintptr_t)act % 16 == 0 && (uintptr_t)w_fc % 16 == 0 && (uintptr_t)workspace % 256 == 0,
                 "fc_emissions: alignment");
    const bool x2 = dtype == LA_F32 && w_x2 && w_x2s && fc_x2_ok(batch * frames, in_dim, vocab);
    const HeadPlan pl = plan_head(batch, frames, in_dim, vocab, max_labels, es, x2);
    LA_CHECK_ARG(workspace_bytes >= pl.total, "fc_emissions: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    void *wg = ws + pl.off_wg;
    float *bg = reinterpret_cast<float *>(ws + pl.off_bg);
    float *raw = reinterpret_cast<float *>(ws + pl.off_raw);
    float2 *partials = reinterpret_cast<float2 *>(ws + pl.off_part);
    const int S = max_labels + 1;
    const int rows = batch * frames;

    // 1. gather
    if (dtype != LA_F32)       // 16-bit rows are copied as raw bits: one instantiation serves bf16 and half
        hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(S, batch), dim3(128), 0, stream, (const bf16_t *)w_fc, b_fc,
                           in_dim, vocab, variant, labels, labels_stride, n_labels, max_labels, (bf16_t *)wg, bg);
    else
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(S, batch), dim3(128), 0, stream, (const float *)w_fc, b_fc,
                           in_dim, vocab, variant, labels, labels_stride, n_labels, max_labels, (float *)wg, bg);
    LA_LAUNCH_CHECK();
    // 2. raw logits of the gathered columns
    int rc = la::gemm_run(dtype, frames, S, in_dim, batch, act, ld_act, (int64_t)frames * ld_act, wg, (int64_t)S * in_dim, raw,
                          S, (int64_t)frames * S, bg, S, nullptr, 0, 0, LA_EPI_BIAS | LA_EPI_OUT_F32, stream);
    if (rc != LA_OK) return rc;
    // 3. row normaliser partials over the full vocabulary
    LseParams lp{rows, vocab, in_dim, act, ld_act, w_fc, b_fc, partials, pl.nparts,
                 variant == LA_VARIANT_CTC ? 1 : 0, variant == LA_VARIANT_CTC ? vocab - 2 : vocab - 1,
                 0, pl.tiles_n, pick_group(in_dim, es, pl.tiles_n)};
    {
        typedef Cfg<2, 2> Small;
        typedef Cfg<4, 3> Big;
        static bool attr_bf16 = false, attr_bf16_big = false, attr_f32 = false, attr_pp = false;
        const int force_tile = la::opts().gemm_tile;
        const bool pp_ok = dtype != LA_F32 && !force_tile && in_dim % 64 == 0 && in_dim >= 128 && (ld_act * 2) % 16 == 0 &&
                           (int64_t)la::cdiv(rows, PP::TM) * la::cdiv(vocab, PP::TN) >= 192;
        if (!pp_ok && !x2) lp.nparts = 2 * pl.tiles_n;      // the 128-column kernels write two strips per tile
        if (x2) {
            static la::DeviceOnce attr_x2;
            if (attr_x2.pending()) {
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS));
                attr_x2.mark();
            }
            void *planes = ws + pl.off_planes;
            float *inv = reinterpret_cast<float *>(ws + pl.off_inv);
            rc = la_split_f16x2(static_cast<const float *>(act), ld_act, rows, in_dim, planes, in_dim, inv, stream_);
            if (rc != LA_OK) return rc;
            lp.A = planes; lp.W = w_x2; lp.sa = inv; lp.sw = w_x2s;
            lp.tiles_m = la::cdiv(rows, PP::TM);
            lp.tiles_n = la::cdiv(vocab, PP::TN);
            lp.group = std::max(1, pick_group(3 * in_dim, 2, la::cdiv(vocab, BN)) / 2);
            la::TimerScope ts("fc_lse_f32", stream);
            hipLaunchKernelGGL(fc_lse_x2_kernel, dim3(lp.tiles_m * lp.tiles_n), dim3(PP::THREADS), PP::LDS, stream, lp);
        } else if (pp_ok) {
            // the strips of the last (partial) 256-column tile that no 128-column tile would have produced must still hold
            // neutral partials: every strip is written by this kernel, masked columns give (-inf, 0)
            if (!attr_pp) {
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_pp_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS));
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_pp_kernel<la::f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS));
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_pp_kernel<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS));
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_pp_kernel<la::f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS));
                attr_pp = true;
            }
            lp.tiles_m = la::cdiv(rows, PP::TM);
            lp.tiles_n = la::cdiv(vocab, PP::TN);
            lp.group = std::max(1, pick_group(in_dim, es, la::cdiv(vocab, BN)) / 2);
            la::TimerScope ts("fc_lse_bf16", stream);
            const bool duo = in_dim % 128 == 0 && in_dim >= 256 && la::opts().gemm_loop != 99;    // (99: the ping-pong loop everywhere)
            if (duo) {
                if (dtype == LA_F16) hipLaunchKernelGGL((fc_lse_pp_kernel<la::f16_t, true>), dim3(lp.tiles_m * lp.tiles_n), dim3(PP::THREADS), PP::LDS, stream, lp);
                else hipLaunchKernelGGL((fc_lse_pp_kernel<bf16_t, true>), dim3(lp.tiles_m * lp.tiles_n), dim3(PP::THREADS), PP::LDS, stream, lp);
            } else if (dtype == LA_F16) hipLaunchKernelGGL(fc_lse_pp_kernel<la::f16_t>, dim3(lp.tiles_m * lp.tiles_n), dim3(PP::THREADS), PP::LDS, stream, lp);
            else hipLaunchKernelGGL(fc_lse_pp_kernel<bf16_t>, dim3(lp.tiles_m * lp.tiles_n), dim3(PP::THREADS), PP::LDS, stream, lp);
        } else if (dtype == LA_BF16 && rows >= 4096 && force_tile == 256) {
            if (!attr_bf16_big) {
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_kernel<bf16_t, Big>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Big::LDS));
                attr_bf16_big = true;
            }
            lp.tiles_m = la::cdiv(rows, Big::TM);
            la::TimerScope ts("fc_lse_bf16", stream);
            hipLaunchKernelGGL((fc_lse_kernel<bf16_t, Big>), dim3(lp.tiles_m * lp.tiles_n), dim3(Big::THREADS), Big::LDS, stream, lp);
        } else if (dtype == LA_BF16 || dtype == LA_F16) {
            if (!attr_bf16) {
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_kernel<bf16_t, Small>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Small::LDS));
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_kernel<la::f16_t, Small>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Small::LDS));
                attr_bf16 = true;
            }
            lp.tiles_m = la::cdiv(rows, Small::TM);
            la::TimerScope ts("fc_lse_bf16", stream);
            if (dtype == LA_F16) hipLaunchKernelGGL((fc_lse_kernel<la::f16_t, Small>), dim3(lp.tiles_m * lp.tiles_n), dim3(Small::THREADS), Small::LDS, stream, lp);
            else hipLaunchKernelGGL((fc_lse_kernel<bf16_t, Small>), dim3(lp.tiles_m * lp.tiles_n), dim3(Small::THREADS), Small::LDS, stream, lp);
        } else {
            if (!attr_f32) {
                LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fc_lse_kernel<float, Small>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Small::LDS));
                attr_f32 = true;
            }
            lp.tiles_m = la::cdiv(rows, Small::TM);
            la::TimerScope ts("fc_lse_f32", stream);
            hipLaunchKernelGGL((fc_lse_kernel<float, Small>), dim3(lp.tiles_m * lp.tiles_n), dim3(Small::THREADS), Small::LDS, stream, lp);
        }
        LA_LAUNCH_CHECK();
    }
    // 4. merge
    hipLaunchKernelGGL(merge_emissions_kernel, dim3(la::cdiv(rows, 4)), dim3(256), 0, stream, partials, lp.nparts, raw,
                       frames, rows, variant, vocab, labels, labels_stride, n_labels, max_labels, em, em_batch_stride,
                       em_row_stride);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_fc_emissions(int32_t dtype, const void *act, int64_t ld_act, const void *w_fc, const float *b_fc,
                               int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab, int32_t variant,
                               const int32_t *labels, int32_t labels_stride, const int32_t *n_labels, int32_t max_labels,
                               float *em, int64_t em_batch_stride, int64_t em_row_stride, void *workspace,
                               size_t workspace_bytes, void *stream) {
    return fc_emissions_impl(dtype, act, ld_act, w_fc, b_fc, nullptr, nullptr, batch, frames, in_dim, vocab, variant, labels, labels_stride, n_labels,
                             max_labels, em, em_batch_stride, em_row_stride, workspace, workspace_bytes, stream);
}

extern "C" int la_fc_emissions_x2_workspace_bytes(int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab, int32_t max_labels, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && frames > 0 && in_dim > 0 && vocab > 0 && max_labels > 0, "fc_emissions_x2_workspace_bytes: bad arguments");
    *bytes = plan_head(batch, frames, in_dim, vocab, max_labels, 4, fc_x2_ok(batch * frames, in_dim, vocab)).total;
    return LA_OK;
}

extern "C" int la_fc_emissions_x2(const float *act, int64_t ld_act, const float *w_fc, const float *b_fc, const void *w_fc_x2, const float *w_fc_x2s,
                                  int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab, int32_t variant,
                                  const int32_t *labels, int32_t labels_stride, const int32_t *n_labels, int32_t max_labels,
                                  float *em, int64_t em_batch_stride, int64_t em_row_stride, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    LA_CHECK_ARG(w_fc_x2 && w_fc_x2s && (uintptr_t)w_fc_x2 % 16 == 0, "fc_emissions_x2: the planes of W_fc and their scales are required (16-byte aligned)");
    return fc_emissions_impl(LA_F32, act, ld_act, w_fc, b_fc, w_fc_x2, w_fc_x2s, batch, frames, in_dim, vocab, variant, labels, labels_stride, n_labels,
                             max_labels, em, em_batch_stride, em_row_stride, workspace, workspace_bytes, stream);
}
