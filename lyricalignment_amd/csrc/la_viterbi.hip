// la_viterbi.hip -- batched forced-alignment DP for gfx950.
//
// Replaces utils/alignment.py:73-119 (run_viterbi_core) and :141-185 (init,
// termination, backtrace, first/last frame per label) of the reference.
//
// Mapping to the hardware (DESIGN.md "Viterbi"): the recurrence reads only row
// j-1, so one workgroup sweeps one utterance row by row with one lane per
// lattice state k (S = 2L+1 states).  For S <= 64 (every Opencpop utterance and
// the 30 s benchmark clips) that is ONE wave64 and the k-1 / k-2 neighbours come
// from DPP wave shifts, no LDS and no barrier in the loop.  Larger lattices
// (long-form songs) use up to 16 waves and a double-buffered f64 row in LDS with
// one barrier per frame.  Scores are float64, comparisons are in the reference's
// order; backpointers are the offsets {0,1,2} packed as two 64-bit ballot masks
// per wave per frame and kept in LDS when they fit (T * 16 B per wave), otherwise
// in the caller's workspace.  The kernel is latency-bound (T dependent steps),
// not bandwidth-bound: emissions are prefetched PF frames ahead into registers.
#include "la_common.h"

namespace {

constexpr double kNeg = -10000000.0;  // utils/alignment.py:144
constexpr int PF = 8;                 // emission prefetch depth (frames)

struct VitParams {
    const float *em;
    int64_t em_bs, em_rs;
    const int32_t *labels;
    int32_t labels_stride;
    const int32_t *n_labels;
    const int32_t *n_frames;
    int32_t max_frames, max_labels;
    int32_t *onset, *offset;
    int32_t out_stride;
    double *final_score;
    int32_t *status;
    unsigned long long *bt_global;  // [batch][max_frames][NW][2] when !bt_in_lds
    int32_t bt_in_lds;
    // run_viterbi_core face (DUMP instantiation only, batch 1): row 0 of dp is READ, rows >= 1 of dp / bt are written
    double *dp_dump;    // [T][S]
    long long *bt_dump; // [T][S]
};

__device__ __forceinline__ double wave_shr1(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    // DPP wave_shr:1 -- lane i receives lane i-1 across the whole wave64 (gfx9 family)
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int NW, bool DPP, bool DUMP = false>
__global__ __launch_bounds__(NW * 64) void viterbi_kernel(VitParams p) {
    static_assert(!DPP || NW == 1, "DPP neighbour exchange is single-wave only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = NW * 64;
    // carve: row exchange [2][NT+2] f64 | on/off [2][max_labels] i32 | bt masks
    double *rowbuf = reinterpret_cast<double *>(smem);
    int32_t *on_s = reinterpret_cast<int32_t *>(smem + 2 * (NT + 2) * sizeof(double));
    const int Lpad = (p.max_labels + 3) & ~3;
    int32_t *off_s = on_s + Lpad;
    unsigned long long *bt_lds = reinterpret_cast<unsigned long long *>(off_s + Lpad);

    const int b = blockIdx.x;
    const int k = threadIdx.x;
    const int wave = k >> 6;
    const int lane = k & 63;
    const int L = p.n_labels[b];
    const int T = p.n_frames[b];
    const int S = 2 * L + 1;

    for (int n = k; n < p.max_labels; n += NT) {
        p.onset[(int64_t)b * p.out_stride + n] = -1;
        p.offset[(int64_t)b * p.out_stride + n] = -1;
        on_s[n] = -1;
        off_s[n] = -1;
    }
    if (L <= 0) {  // reference: IndexError at cur_label[0] (:152)
        if (k == 0) { p.status[b] = LA_EEMPTY; p.final_score[b] = 0.0; }
        return;
    }
    if (T <= 0 || T > p.max_frames || L > p.max_labels || S > NT) {
        if (k == 0) { p.status[b] = LA_EINVAL; p.final_score[b] = 0.0; }
        return;
    }

    unsigned long long *bt = p.bt_in_lds ? bt_lds : p.bt_global + (int64_t)b * p.max_frames * NW * 2;

    const bool valid = k < S;
    const bool odd = (k & 1) != 0;
    const int n = k >> 1;
    const int col = (odd && valid) ? 1 + n : 0;
    bool can_skip = false;  // label[k//2] != label[k//2-1], odd k >= 3 (:104)
    if (odd && valid && k >= 3) {
        const int32_t *lab = p.labels + (int64_t)b * p.labels_stride;
        can_skip = lab[n] != lab[n - 1];
    }
    const float *emb = p.em + (int64_t)b * p.em_bs + col;

    // row 0 (:144-152); the run_viterbi_core face takes row 0 from the caller like the reference does (:73-76)
    double cur = (k <= 1) ? (double)emb[0] : kNeg;
    if (DUMP) cur = valid ? p.dp_dump[k] : kNeg;

    if (!DPP) {
        if (k < 2) { rowbuf[k] = kNeg; rowbuf[NT + 2 + k] = kNeg; }  // slots for k-1, k-2 of states 0,1
    }

    float e_buf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int jj = 1 + i;
        e_buf[i] = jj < T ? emb[(int64_t)jj * p.em_rs] : 0.0f;
    }
    int parity = 0;
    for (int j0 = 1; j0 < T; j0 += PF) {
        float e_cur[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) e_cur[i] = e_buf[i];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int jj = j0 + PF + i;
            e_buf[i] = jj < T ? emb[(int64_t)jj * p.em_rs] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int j = j0 + i;
            if (j >= T) break;
            double p0 = cur, p1, p2;
            if (DPP) {
                p1 = wave_shr1(p0);
                p2 = wave_shr1(p1);
            } else {
                double *rb = rowbuf + parity * (NT + 2);
                rb[k + 2] = p0;
                __syncthreads();
                p1 = rb[k + 1];
                p2 = rb[k];
                parity ^= 1;
            }
            const bool stay = p0 > p1;                                   // strict (:85,:94,:110)
            const bool skip = can_skip && (p2 >= p1) && (p2 >= p0);      // (:104-105)
            int code = skip ? 2 : (stay ? 0 : 1);
            double best = skip ? p2 : (stay ? p0 : p1);
            if (k == 0) { code = 0; best = p0; }                         // (:78-82)
            cur = best + (double)e_cur[i];
            if (DUMP && valid) {
                p.dp_dump[(int64_t)j * S + k] = cur;
                p.bt_dump[(int64_t)j * S + k] = k - code;
            }
            const unsigned long long lo = __ballot(code & 1);
            const unsigned long long hi = __ballot(code >> 1);
            if (lane == 0) {
                unsigned long long *row = bt + ((int64_t)j * NW + wave) * 2;
                row[0] = lo;
                row[1] = hi;
            }
        }
    }

    // termination + backtrace (:157-185) by one thread
    __syncthreads();
    double *fin = rowbuf;
    fin[k] = cur;
    __threadfence_block();
    __syncthreads();
    if (k == 0) {
        int kk = (fin[S - 1] > fin[S - 2]) ? (S - 1) : (S - 2);  // strict '>' (:157)
        p.final_score[b] = fin[kk];
        int knext = -1;
        for (int j = T - 1; j >= 0; --j) {
            if (kk & 1) {
                const int nn = kk >> 1;
                if (kk != knext) off_s[nn] = j + 1;  // last frame in this state + 1
                on_s[nn] = j;                        // keeps decreasing to the first frame
            }
            knext = kk;
            if (j > 0) {
                const unsigned long long *row = bt + ((int64_t)j * NW + (kk >> 6)) * 2;
                const int sh = kk & 63;
                const int code = (int)((row[0] >> sh) & 1ull) | ((int)((row[1] >> sh) & 1ull) << 1);
                kk -= code;
            }
        }
        int st = LA_OK;
        for (int nn = 0; nn < L; ++nn)
            if (on_s[nn] < 0) st = LA_EINFEASIBLE;  // reference: ValueError from list.index (:183)
        p.status[b] = st;
    }
    __syncthreads();
    for (int nn = k; nn < L; nn += NT) {
        p.onset[(int64_t)b * p.out_stride + nn] = on_s[nn];
        p.offset[(int64_t)b * p.out_stride + nn] = off_s[nn];
    }
}


// Lattices beyond 1024 states (whole songs with more than 511 characters; the reference's run_viterbi_core has no limit):
// 1024 threads, each owning R CONSECUTIVE states k = tid*R + r, so 1024*R states per workgroup (R = 2, 4, 8 -> up to 4095
// labels).  Same recurrence, same comparison order, float64; the previous row goes through a double-buffered f64 row in
// LDS (one barrier per frame).  Backpointer masks: [frame][r][wave][2] 64-bit ballots in the caller's workspace.
// Onset / offset go straight to the outputs (no LDS copies: the row buffers take the LDS at R = 8).
template <int R>
__global__ __launch_bounds__(1024) void viterbi_strip_kernel(VitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 1024, NS = NT * R, SPF = 4;
    double *rowbuf = reinterpret_cast<double *>(smem);      // [2][NS + 2]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int L = p.n_labels[b];
    const int T = p.n_frames[b];
    const int S = 2 * L + 1;
    int32_t *on_g = p.onset + (int64_t)b * p.out_stride, *off_g = p.offset + (int64_t)b * p.out_stride;
    for (int n = tid; n < p.max_labels; n += NT) { on_g[n] = -1; off_g[n] = -1; }
    if (L <= 0) {
        if (tid == 0) { p.status[b] = LA_EEMPTY; p.final_score[b] = 0.0; }
        return;
    }
    if (T <= 0 || T > p.max_frames || L > p.max_labels || S > NS) {
        if (tid == 0) { p.status[b] = LA_EINVAL; p.final_score[b] = 0.0; }
        return;
    }
    unsigned long long *bt = p.bt_global + (int64_t)b * p.max_frames * R * 32;
    const int32_t *lab = p.labels + (int64_t)b * p.labels_stride;
    const int k0 = tid * R;
    int col[R];
    bool can_skip[R], valid[R];
    double cur[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = k0 + r, n = k >> 1;
        valid[r] = k < S;
        const bool odd = (k & 1) != 0;
        col[r] = (odd && valid[r]) ? 1 + n : 0;
        can_skip[r] = odd && valid[r] && k >= 3 && lab[n] != lab[n - 1];
    }
    const float *emb = p.em + (int64_t)b * p.em_bs;
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = (k0 + r <= 1) ? (double)emb[col[r]] : kNeg;
    if (tid == 0) { rowbuf[0] = kNeg; rowbuf[1] = kNeg; rowbuf[NS + 2] = kNeg; rowbuf[NS + 3] = kNeg; }

    float e_buf[SPF][R];
#pragma unroll
    for (int i = 0; i < SPF; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) e_buf[i][r] = (1 + i) < T ? emb[(int64_t)(1 + i) * p.em_rs + col[r]] : 0.0f;
    int parity = 0;
    for (int j0 = 1; j0 < T; j0 += SPF) {
        float e_cur[SPF][R];
#pragma unroll
        for (int i = 0; i < SPF; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) e_cur[i][r] = e_buf[i][r];
#pragma unroll
        for (int i = 0; i < SPF; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int jj = j0 + SPF + i;
                e_buf[i][r] = jj < T ? emb[(int64_t)jj * p.em_rs + col[r]] : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < SPF; ++i) {
            const int j = j0 + i;
            if (j >= T) break;
            double *rb = rowbuf + parity * (NS + 2);
            // the two rightmost states of this thread are the k-1 / k-2 neighbours of the next thread's first states
            rb[k0 + 2 + R - 1] = cur[R - 1];
            rb[k0 + 2 + R - 2] = cur[R - 2];
            __syncthreads();
            double prev[R + 2];
            prev[0] = rb[k0];
            prev[1] = rb[k0 + 1];
#pragma unroll
            for (int r = 0; r < R; ++r) prev[r + 2] = cur[r];
            parity ^= 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double p0 = prev[r + 2], p1 = prev[r + 1], p2 = prev[r];
                const bool stay = p0 > p1;
                const bool skip = can_skip[r] && (p2 >= p1) && (p2 >= p0);
                int code = skip ? 2 : (stay ? 0 : 1);
                double best = skip ? p2 : (stay ? p0 : p1);
                if (k0 + r == 0) { code = 0; best = p0; }
                cur[r] = best + (double)e_cur[i][r];
                const unsigned long long lo = __ballot(code & 1);
                const unsigned long long hi = __ballot(code >> 1);
                if (lane == 0) {
                    unsigned long long *row = bt + (((int64_t)j * R + r) * 16 + wave) * 2;
                    row[0] = lo;
                    row[1] = hi;
                }
            }
        }
    }
    // termination + backtrace by one thread; the masks were written by other waves of this workgroup: drain + barrier first
    __syncthreads();
    double *fin = rowbuf;
#pragma unroll
    for (int r = 0; r < R; ++r) fin[k0 + r] = cur[r];
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        int kk = (fin[S - 1] > fin[S - 2]) ? (S - 1) : (S - 2);
        p.final_score[b] = fin[kk];
        int knext = -1, entered = 0;
        for (int j = T - 1; j >= 0; --j) {
            if (kk & 1) {
                const int nn = kk >> 1;
                if (kk != knext) { off_g[nn] = j + 1; ++entered; }
                if (j == 0) on_g[nn] = 0;
            }
            int knew = kk;
            if (j > 0) {
                const int t2 = kk / R, r = kk - t2 * R;
                const unsigned long long *row = bt + (((int64_t)j * R + r) * 16 + (t2 >> 6)) * 2;
                const int sh = t2 & 63;
                const int code = (int)((row[0] >> sh) & 1ull) | ((int)((row[1] >> sh) & 1ull) << 1);
                knew = kk - code;
                if ((kk & 1) && knew != kk) on_g[kk >> 1] = j;   // frame j is the first one in this state
            }
            knext = kk;
            kk = knew;
        }
        // the path is monotone in k, so every odd state is entered at most once: all L labels visited <=> L entries
        p.status[b] = entered == L ? LA_OK : LA_EINFEASIBLE;   // reference: ValueError from list.index (:183)
    }
}

struct VitPlan {
    int strip;   // 0: one lane per state (<= 1024 states); else R = states per thread of the 1024-thread strip kernel
    int nw;
    bool bt_in_lds;
    size_t lds_bytes;
    size_t ws_bytes;
};

constexpr size_t kLdsBudget = 160 * 1024 - 1024;

bool plan_viterbi(int batch, int max_frames, int max_labels, VitPlan *pl) {
    const int S = 2 * max_labels + 1;
    int nw = 1;
    while (nw * 64 < S) nw *= 2;
    pl->strip = 0;
    if (nw > 16) {
        int R = 2;
        while (1024 * R < S) R *= 2;
        if (R > 8) return false;
        pl->strip = R;
        pl->nw = 16;
        pl->bt_in_lds = false;
        pl->lds_bytes = 2 * (size_t)(1024 * R + 2) * sizeof(double);
        pl->ws_bytes = (size_t)batch * max_frames * R * 32 * sizeof(unsigned long long);
        return true;
    }
    const size_t fixed = 2 * (size_t)(nw * 64 + 2) * sizeof(double) + 2 * (size_t)((max_labels + 3) & ~3) * sizeof(int32_t);
    const size_t fixed_al = (fixed + 15) & ~(size_t)15;
    const size_t bt_bytes = (size_t)max_frames * nw * 16;
    pl->nw = nw;
    pl->bt_in_lds = fixed_al + bt_bytes <= kLdsBudget;
    pl->lds_bytes = pl->bt_in_lds ? fixed_al + bt_bytes : fixed_al;
    pl->ws_bytes = pl->bt_in_lds ? 0 : (size_t)batch * bt_bytes;
    return true;
}

template <int NW, bool DPP>
int launch_viterbi(const VitParams &p, const VitPlan &pl, int batch, hipStream_t stream) {
    auto kern = viterbi_kernel<NW, DPP>;
    static la::DeviceOnce attr_once;            // once per instantiation, to the planner's budget (pl.lds_bytes never exceeds it)
    if (pl.lds_bytes > 48 * 1024 && attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        attr_once.mark();
    }
    la::TimerScope ts("viterbi", stream);
    hipLaunchKernelGGL(kern, dim3(batch), dim3(NW * 64), pl.lds_bytes, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

}  // namespace

extern "C" int la_viterbi_workspace_bytes(int32_t batch, int32_t max_frames, int32_t max_labels, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch >= 0 && max_frames > 0 && max_labels > 0, "viterbi_workspace_bytes: bad arguments");
    VitPlan pl;
    if (!plan_viterbi(batch, max_frames, max_labels, &pl)) {
        la::set_error("viterbi: max_labels %d exceeds 4095 (8192 lattice states per workgroup)", max_labels);
        return LA_EUNSUPPORTED;
    }
    *bytes = pl.ws_bytes;
    return LA_OK;
}

extern "C" int la_viterbi_batch(const float *em, int64_t em_batch_stride, int64_t em_row_stride,
                                const int32_t *labels, int32_t labels_stride, const int32_t *n_labels,
                                const int32_t *n_frames, int32_t batch, int32_t max_frames, int32_t max_labels,
                                int32_t *onset, int32_t *offset, int32_t out_stride, double *final_score,
                                int32_t *status, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return LA_OK;
    LA_CHECK_ARG(em && labels && n_labels && n_frames && onset && offset && final_score && status,
                 "viterbi_batch: null pointer");
    LA_CHECK_ARG(batch > 0 && max_frames > 0 && max_labels > 0, "viterbi_batch: bad sizes");
    LA_CHECK_ARG(em_row_stride >= max_labels + 1 && out_stride >= max_labels && labels_stride >= max_labels,
                 "viterbi_batch: strides smaller than max_labels");
    VitPlan pl;
    if (!plan_viterbi(batch, max_frames, max_labels, &pl)) {
        la::set_error("viterbi: max_labels %d exceeds 4095", max_labels);
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(pl.ws_bytes == 0 || (workspace && workspace_bytes >= pl.ws_bytes),
                 "viterbi_batch: workspace too small (%zu < %zu)", workspace_bytes, pl.ws_bytes);
    LA_CHECK_ARG(pl.ws_bytes == 0 || (uintptr_t)workspace % 8 == 0, "viterbi_batch: workspace must be 8-byte aligned");
    VitParams p{em, em_batch_stride, em_row_stride, labels, labels_stride, n_labels, n_frames, max_frames,
                max_labels, onset, offset, out_stride, final_score, status,
                reinterpret_cast<unsigned long long *>(workspace), pl.bt_in_lds ? 1 : 0, nullptr, nullptr};
    const bool no_dpp = !la::opts().viterbi_dpp;
    if (pl.strip) {
        la::TimerScope ts("viterbi", stream);
#define LA_STRIP_CASE(RV)                                                                                                  \
    case RV: {                                                                                                             \
        static la::DeviceOnce attr_once;                                                                                     \
        if (attr_once.pending()) {                                                                                                  \
            LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_strip_kernel<RV>),                           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));                      \
            attr_once.mark();                                                                                              \
        }                                                                                                                  \
        hipLaunchKernelGGL(viterbi_strip_kernel<RV>, dim3(batch), dim3(1024), pl.lds_bytes, stream, p);                    \
        break;                                                                                                             \
    }
        switch (pl.strip) {
            LA_STRIP_CASE(2) LA_STRIP_CASE(4) LA_STRIP_CASE(8)
            default: return LA_EUNSUPPORTED;
        }
#undef LA_STRIP_CASE
        LA_LAUNCH_CHECK();
        return LA_OK;
    }
    switch (pl.nw) {
        case 1: return no_dpp ? launch_viterbi<1, false>(p, pl, batch, stream) : launch_viterbi<1, true>(p, pl, batch, stream);
        case 2: return launch_viterbi<2, false>(p, pl, batch, stream);
        case 4: return launch_viterbi<4, false>(p, pl, batch, stream);
        case 8: return launch_viterbi<8, false>(p, pl, batch, stream);
        case 16: return launch_viterbi<16, false>(p, pl, batch, stream);
    }
    return LA_EUNSUPPORTED;
}

// run_viterbi_core(dp, bt, lp, ls, label) of the reference (utils/alignment.py:73-119) for ONE utterance: same
// kernel, instantiated so that it also writes every dp row (float64) and backpointer (int64 predecessor state).
extern "C" int la_viterbi_core(const float *em, int64_t em_row_stride, const int32_t *labels, int32_t n_labels_host,
                               int32_t n_frames_host, const int32_t *n_labels, const int32_t *n_frames, double *dp,
                               long long *bt, int32_t *scratch_i32, double *scratch_f64, void *workspace,
                               size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LA_CHECK_ARG(em && labels && n_labels && n_frames && dp && bt && scratch_i32 && scratch_f64, "viterbi_core: null pointer");
    LA_CHECK_ARG(n_labels_host > 0 && n_frames_host > 0 && em_row_stride >= n_labels_host + 1, "viterbi_core: bad sizes");
    VitPlan pl;
    if (!plan_viterbi(1, n_frames_host, n_labels_host, &pl) || pl.strip) {
        la::set_error("viterbi_core: more than 511 labels (the dp / bt dump face is one lane per state)");
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(pl.ws_bytes == 0 || (workspace && workspace_bytes >= pl.ws_bytes), "viterbi_core: workspace too small");
    // scratch_i32: onset[L] | offset[L] | status[1]
    VitParams p{em, 0, em_row_stride, labels, n_labels_host, n_labels, n_frames, n_frames_host, n_labels_host,
                scratch_i32, scratch_i32 + n_labels_host, n_labels_host, scratch_f64, scratch_i32 + 2 * n_labels_host,
                reinterpret_cast<unsigned long long *>(workspace), pl.bt_in_lds ? 1 : 0, dp, bt};
#define LA_CORE_CASE(NWV)                                                                                           \
    case NWV: {                                                                                                     \
        auto kern = viterbi_kernel<NWV, false, true>;                                                               \
        if (pl.lds_bytes > 48 * 1024)                                                                               \
            LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)pl.lds_bytes));                                                         \
        hipLaunchKernelGGL(kern, dim3(1), dim3(NWV * 64), pl.lds_bytes, stream, p);                                 \
        break;                                                                                                      \
    }
    switch (pl.nw) {
        LA_CORE_CASE(1) LA_CORE_CASE(2) LA_CORE_CASE(4) LA_CORE_CASE(8) LA_CORE_CASE(16)
        default: return LA_EUNSUPPORTED;
    }
#undef LA_CORE_CASE
    LA_LAUNCH_CHECK();
    return LA_OK;
}
