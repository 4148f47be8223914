// la_loss.hip -- fine-tune losses on the framewise align logits, forward AND gradient w.r.t. the logits
// (train_multitask.py:587-633 of the reference):
//   compute_ce_loss(compute_sil=True):  CrossEntropy over columns 1..V-1 against (frame_label - 1), ignore -100,
//                                       + BCEWithLogits of column V against (frame_label == -100)            (:587-614)
//   compute_ctc_loss:                   log_softmax over columns 0..V-1, F.ctc_loss(blank=0, reduction='mean',
//                                       zero_infinity=False), input_lengths = T, target_lengths = #labels      (:616-633)
// logits are [B][T][ldl >= V+1] f32 (V = 21128 word columns incl. blank, column V = silence logit).
//
// Kernels (all HBM- or latency-bound, none GEMM-shaped):
//   row_stats        one workgroup per (b,t) row: max / sum-exp over [0,V) and over [1,V)  -> 2 log-normalisers, and the
//                    CE / BCE loss terms of the row (block-reduced, one atomic per workgroup)
//   ctc_lattice      one workgroup per utterance, one lane per extended-label state s (S = 2L+1): alpha sweep forward
//                    (stored), beta sweep backward, nll, and the per-(t,s) occupancy exp(alpha+beta+nll-lp) which is
//                    scattered into the gradient (the alpha/beta lattice north_star names; same row-by-row wave sweep
//                    as the Viterbi kernel with max replaced by log-sum-exp)
//   dense_grad       one workgroup per row: dlogits = scale * ( w_ctc_b * softmax_[0,V) + valid/n_valid * softmax_[1,V)
//                    - one-hot terms ), column V gets the BCE gradient
#include "la_common.h"

namespace {

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// log(exp(a) + exp(b)), -inf safe.  float64: path scores reach -T*log(V) ~ -1e4, where float32 has ~1e-3 absolute
// resolution and the occupancies exp(alpha+beta+nll-lp) lose 2-3 digits over a 1500-step recursion (measured 0.4 %).
__device__ __forceinline__ double log_add(double a, double b) {
    const double m = fmax(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + log1p(exp(fmin(a, b) - m));
}

struct LossAcc {  // device accumulators (double): [0] sum CE, [1] #valid frames, [2] sum BCE, [3] sum_b nll_b / L_b, [4] #inf
    double v[8];
};

// ---- row statistics + CE / BCE terms --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_stats_kernel(const float *logits, int64_t ldl, int rows, int V,
                                                        const int32_t *frame_labels, float *lse_all, float *lse_ce,
                                                        LossAcc *acc) {
    __shared__ float red[12];
    const int row = blockIdx.x;
    const float *x = logits + (int64_t)row * ldl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int c = 1 + tid; c < V; c += 256) m = fmaxf(m, x[c]);
    m = wave_max_f(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    const float m1 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));  // max over [1,V)
    const float x0 = x[0];
    const float m0 = fmaxf(m1, x0);                                          // max over [0,V)
    float s = 0.f;
    for (int c = 1 + tid; c < V; c += 256) s += expf(x[c] - m1);
    s = wave_sum_f(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    if (tid == 0) {
        const float s1 = (red[4] + red[5]) + (red[6] + red[7]);
        const float l_ce = m1 + logf(s1);
        const float l_all = m0 + logf(s1 * expf(m1 - m0) + expf(x0 - m0));
        lse_ce[row] = l_ce;
        lse_all[row] = l_all;
        if (frame_labels) {
            const int lab = frame_labels[row];
            const float xs = x[V];
            const float y = lab == -100 ? 1.f : 0.f;
            const float bce = fmaxf(xs, 0.f) - xs * y + log1pf(expf(-fabsf(xs)));   // BCEWithLogitsLoss, stable form
            atomicAdd(&acc->v[2], (double)bce);
            if (lab != -100) {
                const int c = lab;  // class id k (>= 1): reference shifts labels by -1 and slices columns 1.., i.e. column k
                if (c >= 1 && c < V) {
                    atomicAdd(&acc->v[0], (double)(l_ce - x[c]));
                    atomicAdd(&acc->v[1], 1.0);
                }
            }
        }
    }
}

// ---- CTC alpha / beta lattice -----------------------------------------------------------------------------------
// One workgroup (NT threads, NT >= S) per utterance.  lp_t(s) = logits[t][ext(s)] - lse_all[t], ext(s) = 0 for even s,
// label[s/2] for odd s.  alpha rows are stored in the workspace for the backward sweep.
template <int NT>
__global__ __launch_bounds__(NT) void ctc_lattice_kernel(const float *logits, int64_t ld_b, int64_t ldl, int T, int V,
                                                         const float *lse_all, const int32_t *labels, int labels_stride,
                                                         const int32_t *n_labels, double *alpha_ws, int S_pad,
                                                         float *dlogits, int64_t ldd_b, int64_t ldd, float scale,
                                                         int batch, LossAcc *acc, float *nll_out, int reuse_alpha) {
    __shared__ double rowbuf[2][NT + 2];
    __shared__ double fin[2];
    const int b = blockIdx.x, s = threadIdx.x;
    const int L = n_labels[b];
    const int S = 2 * L + 1;
    const bool valid = s < S && L > 0;
    const int32_t *lab = labels + (int64_t)b * labels_stride;
    const int cls = (valid && (s & 1)) ? lab[s >> 1] : 0;
    const bool can_skip = valid && (s & 1) && s >= 3 && lab[s >> 1] != lab[(s >> 1) - 1];
    // backward-direction skip: state s may go to s+2 iff ext(s+2) != blank and != ext(s)
    const bool can_skip_fwd_from = valid && (s & 1) && (s + 2 < S) && lab[(s >> 1) + 1] != lab[s >> 1];
    const float *xb = logits + (int64_t)b * ld_b;
    const float *lseb = lse_all + (int64_t)b * T;
    double *aw = alpha_ws + (int64_t)b * T * S_pad;
    const bool cls_ok = cls >= 0 && cls < V;
    if (L <= 0 || S > NT) {  // torch: zero-length targets give nll = -sum lp(blank); not used by the reference's data
        if (s == 0) { nll_out[b] = 0.f; }
        return;
    }
    if (s < 2) { rowbuf[0][s] = -INFINITY; rowbuf[1][s] = -INFINITY; }
    int par = 0;
    double nll;
    if (reuse_alpha) {
        // gradient launch after the loss launch: the alpha rows (and with them log p(labels)) are in the workspace already
        const double ll = S > 1 ? log_add(aw[(int64_t)(T - 1) * S_pad + S - 1], aw[(int64_t)(T - 1) * S_pad + S - 2])
                                : aw[(int64_t)(T - 1) * S_pad + S - 1];
        nll = -ll;
    } else {
        // ---- alpha ----
        double a = -INFINITY;
        {
            const double lp = valid && cls_ok ? (double)xb[cls] - (double)lseb[0] : -INFINITY;
            if (s <= 1 && valid) a = lp;
            if (valid) aw[s] = a;
        }
        for (int t = 1; t < T; ++t) {
            rowbuf[par][s + 2] = a;
            __syncthreads();
            const double a1 = rowbuf[par][s + 1], a2 = rowbuf[par][s];
            par ^= 1;
            double acc3 = log_add(a, a1);
            if (can_skip) acc3 = log_add(acc3, a2);
            const double lp = valid && cls_ok ? (double)xb[(int64_t)t * ldl + cls] - (double)lseb[t] : -INFINITY;
            a = valid ? acc3 + lp : -INFINITY;
            if (valid) aw[(int64_t)t * S_pad + s] = a;
        }
        __syncthreads();
        if (s == S - 1) fin[0] = a;
        if (s == S - 2) fin[1] = a;
        __syncthreads();
        const double ll = S > 1 ? log_add(fin[0], fin[1]) : fin[0];
        nll = -ll;
        if (s == 0) {
            nll_out[b] = (float)nll;
            if (isinf(nll)) atomicAdd(&acc->v[4], 1.0);
            atomicAdd(&acc->v[3], nll / (double)L);
        }
    }
    if (!dlogits) return;
    // ---- beta + occupancy scatter ----
    // beta_{T-1}(s) = lp_{T-1}(s) for s in {S-1, S-2}; beta_t(s) = lp_t(s) + logsumexp(beta_{t+1}(s), beta_{t+1}(s+1), [beta_{t+1}(s+2)])
    // gradient of mean_b(nll_b / L_b): d/dlogits[t][c] = w * ( softmax[t][c] - sum_{s: ext(s)=c} exp(alpha_t(s) + beta_t(s) + nll - lp_t(s)) )
    // the softmax part is written by dense_grad_kernel; here the occupancy part is subtracted with float atomics.
    const float w = scale / ((float)batch * (float)L);
    float *db = dlogits + (int64_t)b * ldd_b;
    __syncthreads();
    if (s < 2) { rowbuf[0][NT + s] = -INFINITY; rowbuf[1][NT + s] = -INFINITY; }  // slots for s+1, s+2 beyond the top
    double be = -INFINITY;
    par = 0;
    for (int t = T - 1; t >= 0; --t) {
        const double lp = valid && cls_ok ? (double)xb[(int64_t)t * ldl + cls] - (double)lseb[t] : -INFINITY;
        if (t == T - 1) {
            be = (valid && (s == S - 1 || s == S - 2)) ? lp : -INFINITY;
        } else {
            rowbuf[par][s] = be;
            __syncthreads();
            const double b1 = rowbuf[par][s + 1], b2 = rowbuf[par][s + 2];
            par ^= 1;
            double acc3 = log_add(be, b1);
            if (can_skip_fwd_from) acc3 = log_add(acc3, b2);
            be = valid ? acc3 + lp : -INFINITY;
        }
        if (valid && cls_ok && !isinf(nll)) {
            const double al = aw[(int64_t)t * S_pad + s];
            const float occ = (float)exp(al + be + nll - lp);   // alpha and beta both include lp_t(s) once
            if (occ != 0.f) atomicAdd(&db[(int64_t)t * ldd + cls], -w * occ);
        }
    }
}

// ---- the same lattice for S <= 64: one WAVE per utterance --------------------------------------------------------------------------------
// The recursion is a dependency chain of T steps (1500 for a 30 s clip) per utterance, and a fine-tune micro-batch holds two CTC clips: the
// launch is two waves on the whole chip and its time is the chain's latency -- 0.78 us per step with the workgroup form above (LDS exchange
// + barrier, float64 exp / log1p: 37 ms of a 450 ms optimizer step).  Here a step is: neighbours by DPP wave shifts (no LDS), ONE three-way
// log-sum-exp with its maximum in float64 and the correction log(sum exp(x - max)) in float32 -- the correction lies in [0, ln 3], so its
// float32 rounding is <= 1e-7 ABSOLUTE per step (the path scores themselves, ~ -1e4, stay float64: the reason the lattice is float64 at
// all) -- and the emissions / stored alpha rows prefetched a block of 8 steps ahead of the chain.
__device__ __forceinline__ double wave_shr1(double x, double fill) {      // lane i <- lane i-1, lane 0 <- fill
    const int lo = __double2loint(x), hi = __double2hiint(x), flo = __double2loint(fill), fhi = __double2hiint(fill);
    return __hiloint2double(__builtin_amdgcn_update_dpp(fhi, hi, 0x138, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(flo, lo, 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_shl1(double x, double fill) {      // lane i <- lane i+1, lane 63 <- fill
    const int lo = __double2loint(x), hi = __double2hiint(x), flo = __double2loint(fill), fhi = __double2hiint(fill);
    return __hiloint2double(__builtin_amdgcn_update_dpp(fhi, hi, 0x130, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(flo, lo, 0x130, 0xf, 0xf, false));
}
// log(exp(a) + exp(b) + exp(c)), -inf safe
__device__ __forceinline__ double log_add3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    if (m == -INFINITY) return -INFINITY;
    const float sum = __expf((float)(a - m)) + __expf((float)(b - m)) + __expf((float)(c - m));
    return m + (double)__logf(sum);
}

__global__ __launch_bounds__(64) void ctc_lattice_wave_kernel(const float *logits, int64_t ld_b, int64_t ldl, int T, int V, const float *lse_all,
                                                              const int32_t *labels, int labels_stride, const int32_t *n_labels, double *alpha_ws,
                                                              int S_pad, float *dlogits, int64_t ldd_b, int64_t ldd, float scale, int batch,
                                                              LossAcc *acc, float *nll_out, int reuse_alpha) {
    constexpr int U = 8;                  // steps per prefetch block
    const int b = blockIdx.x, s = threadIdx.x;
    const int L = n_labels[b];
    const int S = 2 * L + 1;
    if (L <= 0 || S > 64) {
        if (s == 0) nll_out[b] = 0.f;
        return;
    }
    const bool valid = s < S;
    const int32_t *lab = labels + (int64_t)b * labels_stride;
    const int cls = (valid && (s & 1)) ? lab[s >> 1] : 0;
    const bool can_skip = valid && (s & 1) && s >= 3 && lab[s >> 1] != lab[(s >> 1) - 1];
    const bool can_skip_fwd_from = valid && (s & 1) && (s + 2 < S) && lab[(s >> 1) + 1] != lab[s >> 1];
    const bool live = valid && cls >= 0 && cls < V;        // lanes with an emission; the others carry -inf
    const float *xb = logits + (int64_t)b * ld_b + (live ? cls : 0);
    const float *lseb = lse_all + (int64_t)b * T;
    double *aw = alpha_ws + (int64_t)b * T * S_pad + (valid ? s : 0);
    const double NEG = -INFINITY;
    double nll;
    if (reuse_alpha) {
        const double a_last = aw[(int64_t)(T - 1) * S_pad];
        const double f0 = __shfl(a_last, S - 1), f1 = S > 1 ? __shfl(a_last, S - 2) : NEG;
        nll = -log_add(f0, f1);
    } else {
        // ---- alpha ----
        double a = (s <= 1 && live) ? (double)xb[0] - (double)lseb[0] : NEG;
        if (valid) aw[0] = a;
        float xv[U], lv[U];
        auto fetch = [&](int t0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = min(t0 + u, T - 1);
                xv[u] = xb[(int64_t)t * ldl];
                lv[u] = lseb[t];
            }
        };
        fetch(1);
        for (int t0 = 1; t0 < T; t0 += U) {
            float xc[U], lc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { xc[u] = xv[u]; lc[u] = lv[u]; }
            if (t0 + U < T) fetch(t0 + U);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u;
                if (t < T) {                                          // wave-uniform
                    const double a1 = wave_shr1(a, NEG);
                    double a2 = wave_shr1(a1, NEG);
                    if (!can_skip) a2 = NEG;
                    const double lp = (double)xc[u] - (double)lc[u];
                    a = live ? log_add3(a, a1, a2) + lp : NEG;
                    if (valid) aw[(int64_t)t * S_pad] = a;
                }
            }
        }
        const double f0 = __shfl(a, S - 1), f1 = S > 1 ? __shfl(a, S - 2) : NEG;
        nll = -log_add(f0, f1);
        if (s == 0) {
            nll_out[b] = (float)nll;
            if (isinf(nll)) atomicAdd(&acc->v[4], 1.0);
            atomicAdd(&acc->v[3], nll / (double)L);
        }
    }
    if (!dlogits) return;
    // ---- beta + occupancy scatter (see ctc_lattice_kernel) ----
    const float w = scale / ((float)batch * (float)L);
    float *db = dlogits + (int64_t)b * ldd_b + (live ? cls : 0);
    const bool scatter = live && !isinf(nll);
    double be = NEG;
    float xv[U], lv[U];
    double av[U];
    auto fetch = [&](int t0) {                                        // steps t0, t0 - 1, ..., t0 - U + 1
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = max(t0 - u, 0);
            xv[u] = xb[(int64_t)t * ldl];
            lv[u] = lseb[t];
            av[u] = aw[(int64_t)t * S_pad];
        }
    };
    fetch(T - 1);
    for (int t0 = T - 1; t0 >= 0; t0 -= U) {
        float xc[U], lc[U];
        double ac[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xc[u] = xv[u]; lc[u] = lv[u]; ac[u] = av[u]; }
        if (t0 - U >= 0) fetch(t0 - U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 - u;
            if (t >= 0) {                                             // wave-uniform
                const double lp = (double)xc[u] - (double)lc[u];
                if (t == T - 1) {
                    be = (live && (s == S - 1 || s == S - 2)) ? lp : NEG;
                } else {
                    const double b1 = wave_shl1(be, NEG);
                    double b2 = wave_shl1(b1, NEG);
                    if (!can_skip_fwd_from) b2 = NEG;
                    be = live ? log_add3(be, b1, b2) + lp : NEG;
                }
                if (scatter) {
                    const float occ = __expf((float)(ac[u] + be + nll - lp));   // alpha and beta both include lp_t(s) once
                    if (occ != 0.f) atomicAdd(&db[(int64_t)t * ldd], -w * occ);
                }
            }
        }
    }
}

// ---- dense gradient ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_grad_kernel(const float *logits, int64_t ldl, int rows, int T, int V,
                                                         const float *lse_all, const float *lse_ce,
                                                         const int32_t *frame_labels, const int32_t *n_labels,
                                                         const float *nll, const LossAcc *acc, int use_ce, int use_ctc,
                                                         float scale, int batch, float *dlogits, int64_t ldd) {
    const int row = blockIdx.x;
    const int b = row / T;
    const float *x = logits + (int64_t)row * ldl;
    float *d = dlogits + (int64_t)row * ldd;
    const float la = lse_all[row], lc = lse_ce[row];
    float w_ctc = 0.f;
    if (use_ctc) {
        const int L = n_labels[b];
        if (L > 0 && !isinf(nll[b])) w_ctc = scale / ((float)batch * (float)L);
    }
    float w_ce = 0.f;
    int lab = -100;
    if (use_ce) {
        lab = frame_labels[row];
        const float n_valid = (float)acc->v[1];
        if (lab != -100 && lab >= 1 && lab < V && n_valid > 0.f) w_ce = scale / n_valid;
    }
    for (int c = threadIdx.x; c < V; c += 256) {
        const float xc = x[c];
        float g = w_ctc * expf(xc - la);
        if (c >= 1) {
            g += w_ce * expf(xc - lc);
            if (c == lab) g -= w_ce;
        }
        d[c] = g;
    }
    if (threadIdx.x == 0) {
        float g = 0.f;
        if (use_ce) {
            const float xs = x[V];
            const float y = lab == -100 ? 1.f : 0.f;
            g = scale * (1.0f / (1.0f + expf(-xs)) - y) / (float)rows;
        }
        d[V] = g;
    }
}

__global__ void finish_losses_kernel(const LossAcc *acc, int rows, int batch, float *out) {
    // out[0] = word CE (mean over valid frames), out[1] = silence BCE (mean over all frames), out[2] = CTC (mean_b nll_b/L_b)
    out[0] = acc->v[1] > 0 ? (float)(acc->v[0] / acc->v[1]) : NAN;   // torch: mean over zero elements is nan
    out[1] = (float)(acc->v[2] / (double)rows);
    out[2] = acc->v[4] > 0 ? INFINITY : (float)(acc->v[3] / (double)batch);
}

}  // namespace

extern "C" int la_multitask_loss_workspace_bytes(int32_t batch, int32_t frames, int32_t max_labels, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && frames > 0 && max_labels > 0, "multitask_loss_workspace_bytes: bad arguments");
    const int64_t rows = (int64_t)batch * frames;
    const int S_pad = (2 * max_labels + 1 + 3) & ~3;
    *bytes = (size_t)(256 + la::round_up(rows * 4, 256) * 2 + la::round_up((int64_t)batch * 4, 256) + la::round_up(rows * S_pad * 8, 256));
    return LA_OK;
}

extern "C" int la_multitask_loss(const float *logits, int64_t batch_stride, int64_t row_stride, int32_t batch, int32_t frames,
                                 int32_t vocab, const int32_t *frame_labels, const int32_t *ctc_labels, int32_t labels_stride,
                                 const int32_t *n_labels, int32_t max_labels, int32_t use_ce, int32_t use_ctc, float scale,
                                 float *losses, float *dlogits, int64_t d_batch_stride, int64_t d_row_stride, void *workspace,
                                 size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LA_CHECK_ARG(logits && losses && workspace, "multitask_loss: null pointer");
    LA_CHECK_ARG(batch > 0 && frames > 0 && vocab >= 3, "multitask_loss: bad sizes");
    LA_CHECK_ARG(batch_stride == (int64_t)frames * row_stride && row_stride >= vocab + 1, "multitask_loss: logits must be [B][T][>= V+1] with dense rows");
    LA_CHECK_ARG(!use_ce || frame_labels, "multitask_loss: CE requested without frame labels");
    LA_CHECK_ARG(!use_ctc || (ctc_labels && n_labels && max_labels > 0 && labels_stride >= max_labels), "multitask_loss: CTC requested without labels");
    LA_CHECK_ARG(!dlogits || (d_batch_stride == (int64_t)frames * d_row_stride && d_row_stride >= vocab + 1), "multitask_loss: dlogits layout");
    LA_CHECK_ARG(2 * max_labels + 1 <= 1024, "multitask_loss: more than 511 labels");
    size_t need = 0;
    la_multitask_loss_workspace_bytes(batch, frames, max_labels > 0 ? max_labels : 1, &need);
    LA_CHECK_ARG(workspace_bytes >= need && (uintptr_t)workspace % 256 == 0, "multitask_loss: workspace too small or misaligned");
    const int rows = batch * frames;
    const int S_pad = (2 * max_labels + 1 + 3) & ~3;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    LossAcc *acc = reinterpret_cast<LossAcc *>(ws);
    float *lse_all = reinterpret_cast<float *>(ws + 256);
    float *lse_ce = lse_all + la::round_up(rows, 64);
    float *nll = lse_ce + la::round_up(rows, 64);
    double *alpha_ws = reinterpret_cast<double *>(nll + la::round_up(batch, 64));
    LA_HIP(hipMemsetAsync(acc, 0, 256, stream));
    la::TimerScope ts("multitask_loss", stream);
    hipLaunchKernelGGL(row_stats_kernel, dim3(rows), dim3(256), 0, stream, logits, row_stride, rows, vocab,
                       use_ce ? frame_labels : nullptr, lse_all, lse_ce, acc);
    LA_LAUNCH_CHECK();
    if (dlogits) {
        // dense part first (it needs nll only to zero the CTC weight of infeasible utterances -> run the lattice's alpha
        // pass first when CTC is on); order: lattice(alpha, nll) is inside ctc_lattice_kernel, so: zero nll, lattice
        // WITHOUT gradient, dense, lattice scatter.  Two lattice launches keep every kernel simple; the lattice is tiny.
    }
    const int S = 2 * max_labels + 1;
    auto launch_lattice = [&](float *dl) {
        if (!use_ctc) return;
#define LA_CTC_CASE(NTV)                                                                                                   \
    hipLaunchKernelGGL((ctc_lattice_kernel<NTV>), dim3(batch), dim3(NTV), 0, stream, logits, batch_stride, row_stride, frames,  \
                       vocab, lse_all, ctc_labels, labels_stride, n_labels, alpha_ws, S_pad, dl, d_batch_stride, d_row_stride,   \
                       scale, batch, dl ? acc + 1 : acc, nll, dl ? 1 : 0)
        if (S <= 64)
            hipLaunchKernelGGL(ctc_lattice_wave_kernel, dim3(batch), dim3(64), 0, stream, logits, batch_stride, row_stride, frames, vocab, lse_all,
                               ctc_labels, labels_stride, n_labels, alpha_ws, S_pad, dl, d_batch_stride, d_row_stride, scale, batch,
                               dl ? acc + 1 : acc, nll, dl ? 1 : 0);
        else if (S <= 128) LA_CTC_CASE(128);
        else if (S <= 256) LA_CTC_CASE(256);
        else if (S <= 512) LA_CTC_CASE(512);
        else LA_CTC_CASE(1024);
#undef LA_CTC_CASE
    };
    if (use_ctc) {
        launch_lattice(nullptr);   // alpha sweep: nll per utterance + loss accumulators
        LA_LAUNCH_CHECK();
    } else {
        LA_HIP(hipMemsetAsync(nll, 0, (size_t)batch * 4, stream));
    }
    if (dlogits) {
        hipLaunchKernelGGL(dense_grad_kernel, dim3(rows), dim3(256), 0, stream, logits, row_stride, rows, frames, vocab, lse_all,
                           lse_ce, frame_labels, n_labels, nll, acc, use_ce, use_ctc, scale, batch, dlogits, d_row_stride);
        LA_LAUNCH_CHECK();
        if (use_ctc) {
            launch_lattice(dlogits);   // beta sweep on the stored alpha rows, occupancies subtracted from the dense softmax part
            LA_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(finish_losses_kernel, dim3(1), dim3(1), 0, stream, acc, rows, batch, losses);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
