// la_attention_bwd.hip -- fused float32 attention backward (training path; whisper/model.py MultiHeadAttention.qkv_attention under
// autograd, reached from train_multitask.py:325-326 `loss.backward()`), head_dim 64.
//
// Round 1-2 composed the backward per clip out of batched float32 GEMMs over MATERIALISED [heads][Tq][Tk] score tiles (S, P, dP,
// dS: 144 MB each at T = 1500) plus two softmax passes: ~1.3 GB of HBM traffic and 7 launches per clip and layer -- 55 % of the
// fused fine-tune step.  Here nothing of size Tq x Tk leaves the CU (flash-attention style, two sweeps so that no gradient needs
// an atomic):
//   attn_stats_kernel   per query row: lse = log sum_j exp(s_ij) (scores recomputed tile by tile, online max / sum) and
//                       D_i = sum_d dO_id O_id  (= sum_j P_ij dP_ij)
//   attn_bwd_kv_kernel  one workgroup per 64-key block, sweeping the query blocks:  dV_j += P^T dO,  dK_j += dS^T Q
//   attn_bwd_q_kernel   one workgroup per 64-query block, sweeping the key blocks:  dQ_i += dS K
// with  P = exp(S - lse),  dP = dO V^T,  dS = P o (dP - D)  recomputed per 64 x 64 tile from LDS-resident operand tiles.
// All five products are v_mfma_f32_16x16x4_f32 with both operands read from LDS tiles [64][64] f32 (row pitch 68 floats):
// K-contiguous operands as 16-byte fragments, operands whose contraction index is the tile's ROW index (P^T, dS^T, dO^T, Q^T,
// K^T -- never materialised) as four 4-byte reads down a column.  q is expected pre-scaled (head_dim^-0.5 folded in), as the
// forward kernels take it; dq is the gradient with respect to that pre-scaled q.
#include "la_gemm_core.h"

namespace {

using la::gemm::f32x4;
using la::gemm::Mma;

constexpr int BT = 64;          // tile edge: queries per block, keys per block, head_dim
constexpr int PITCH = 68;       // floats per LDS tile row (272 B: rows start 16 B apart modulo 128 B)
constexpr int TILE = BT * PITCH;
constexpr float kLog2e = 1.4426950408889634f;

struct BwdParams {
    const float *q, *k, *v, *o, *dout;
    float *dq, *dk, *dv;
    int64_t ld_q, ld_kv, ld_o, ld_do, ld_dq, ld_dkv;
    int B, Tq, Tk, H, causal;
    float *lse, *dvec;          // [B][H][Tq]
    int have_lse;               // lse was handed over by the forward kernel (la_attention_lse_f32): the statistics sweep only takes D
};

// Workgroup shape of the two sweep kernels: NT threads = NT / 64 waves as 2 (row halves of 32) x WN (column strips of 64 / WN);
// a wave holds 2 x NI accumulator tiles of 16 x 16 per product.  8 waves (two per SIMD, one workgroup per CU: the kv sweep's
// six tiles are 102 KB of LDS) hide the fragment-read latency that 4 waves leave exposed: 10.1 -> 7.65 ms per layer for 16 clips
// of Whisper-medium (then 6.0 with the next block's tiles fetched into registers under the current block's products).
constexpr int NT = 512, WN = NT / 128, NI = 4 / WN;

// rows row0 .. row0 + 63 of a row-major matrix (64 columns from `src`), zero beyond `limit`, into an LDS tile
template <int NTH>
__device__ __forceinline__ void load_tile(float *T, const float *src, int64_t ld, int row0, int limit, int tid) {
#pragma unroll
    for (int it = 0; it < 1024 / NTH; ++it) {
        const int idx = tid + it * NTH, row = idx >> 4, c4 = idx & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < limit) v = *reinterpret_cast<const float4 *>(src + (int64_t)(row0 + row) * ld + c4 * 4);
        *reinterpret_cast<float4 *>(T + row * PITCH + c4 * 4) = v;
    }
}
// the same in two halves, so that the global loads of the NEXT tile are in flight while the current one is computed on
template <int NTH>
__device__ __forceinline__ void fetch_tile(float4 (&v)[1024 / NTH], const float *src, int64_t ld, int row0, int limit, int tid) {
#pragma unroll
    for (int it = 0; it < 1024 / NTH; ++it) {
        const int idx = tid + it * NTH, row = idx >> 4, c4 = idx & 15;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < limit) v[it] = *reinterpret_cast<const float4 *>(src + (int64_t)(row0 + row) * ld + c4 * 4);
    }
}
template <int NTH>
__device__ __forceinline__ void put_tile(float *T, const float4 (&v)[1024 / NTH], int tid) {
#pragma unroll
    for (int it = 0; it < 1024 / NTH; ++it) {
        const int idx = tid + it * NTH, row = idx >> 4, c4 = idx & 15;
        *reinterpret_cast<float4 *>(T + row * PITCH + c4 * 4) = v[it];
    }
}
__device__ __forceinline__ uint4 frag_n(const float *T, int row, int k0) {      // 4 consecutive k of one row
    return *reinterpret_cast<const uint4 *>(T + row * PITCH + k0);
}
__device__ __forceinline__ uint4 frag_t(const float *T, int k0, int col) {      // rows k0 .. k0 + 3 of one column
    uint4 v;
    v.x = __float_as_uint(T[(k0 + 0) * PITCH + col]);
    v.y = __float_as_uint(T[(k0 + 1) * PITCH + col]);
    v.z = __float_as_uint(T[(k0 + 2) * PITCH + col]);
    v.w = __float_as_uint(T[(k0 + 3) * PITCH + col]);
    return v;
}

// acc[mi][ni][j] += sum_k A[m][k] W[n][k],  m = 32 wm + 16 mi + r,  n = 16 NI wn + 16 ni + 4 q + j,  k = 0 .. 63.
// TA / TW: that operand's tile is stored [k][m] / [k][n] (the contraction index is the tile's row index).
template <bool TA, bool TW>
__device__ __forceinline__ void mma64(f32x4 (&acc)[2][NI], const float *At, const float *Wt, int wm, int wn, int r, int q) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int k0 = 16 * ks + 4 * q;
        uint4 af[2], wf[NI];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = 32 * wm + 16 * i + r;
            af[i] = TA ? frag_t(At, k0, m) : frag_n(At, m, k0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = 16 * NI * wn + 16 * i + r;
            wf[i] = TW ? frag_t(Wt, k0, n) : frag_n(Wt, n, k0);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) Mma<float>::run(wf[ni], af[mi], acc[mi][ni]);
    }
}
__device__ __forceinline__ void zero(f32x4 (&a)[2][NI]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// P and dS of one 64 x 64 tile in the accumulator layout (m = query, n = key), from the Q / dO / K / V tiles in LDS
__device__ __forceinline__ void tile_p_ds(const BwdParams &p, const float *Qt, const float *dOt, const float *Kt, const float *Vt, int i0, int j0,
                                          const float (&lse)[2], const float (&dv)[2], int wm, int wn, int r, int q,
                                          f32x4 (&P)[2][NI], f32x4 (&dS)[2][NI]) {
    zero(P);
    zero(dS);
    mma64<false, false>(P, Qt, Kt, wm, wn, r, q);          // S = q k^T
    mma64<false, false>(dS, dOt, Vt, wm, wn, r, q);        // dP = dO v^T
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int qg = i0 + 32 * wm + 16 * mi + r;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kg = j0 + 16 * NI * wn + 16 * ni + 4 * q + j;
                const bool ok = kg < p.Tk && qg < p.Tq && (!p.causal || kg <= qg);
                const float pv = ok ? __builtin_amdgcn_exp2f((P[mi][ni][j] - lse[mi]) * kLog2e) : 0.f;
                P[mi][ni][j] = pv;
                dS[mi][ni][j] = pv * (dS[mi][ni][j] - dv[mi]);
            }
    }
}
__device__ __forceinline__ void store_acc_tile(float *T, const f32x4 (&a)[2][NI], int wm, int wn, int r, int q) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            *reinterpret_cast<f32x4 *>(T + (32 * wm + 16 * mi + r) * PITCH + 16 * NI * wn + 16 * ni + 4 * q) = a[mi][ni];
}
__device__ __forceinline__ void store_acc_global(float *dst, int64_t ld, int row0, int limit, const f32x4 (&a)[2][NI], int wm, int wn, int r, int q) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int row = row0 + 32 * wm + 16 * mi + r;
        if (row < limit) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                *reinterpret_cast<f32x4 *>(dst + (int64_t)row * ld + 16 * NI * wn + 16 * ni + 4 * q) = a[mi][ni];
        }
    }
}

// ---- lse, D per query row ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_stats_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Qt = lds, *Kt = lds + TILE;
    const int i0 = blockIdx.x * BT, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
    const float *qb = p.q + (int64_t)b * p.Tq * p.ld_q + h * 64;
    const float *kb = p.k + (int64_t)b * p.Tk * p.ld_kv + h * 64;
    const int row = 16 * w + r, qg = i0 + row;       // this lane's query row (shared by the 4 lanes q = 0..3)
    float m_run = -INFINITY, l_run = 0.f;
    int nkb = (p.Tk + BT - 1) / BT;
    if (p.have_lse) nkb = 0;
    else load_tile<256>(Qt, qb, p.ld_q, i0, p.Tq, tid);
    if (p.causal) nkb = min(nkb, (min(p.Tq, i0 + BT) - 1) / BT + 1);
    for (int jb = 0; jb < nkb; ++jb) {
        const int j0 = jb * BT;
        __syncthreads();
        load_tile<256>(Kt, kb, p.ld_kv, j0, p.Tk, tid);
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) s[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k0 = 16 * ks + 4 * q;
            const uint4 af = frag_n(Qt, row, k0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) Mma<float>::run(frag_n(Kt, 16 * ni + r, k0), af, s[ni]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kg = j0 + 16 * ni + 4 * q + j;
                if (kg >= p.Tk || (p.causal && kg > qg)) s[ni][j] = -INFINITY;
                mx = fmaxf(mx, s[ni][j]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        float ps = 0.f;
        if (m_new > -INFINITY) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) ps += __builtin_amdgcn_exp2f((s[ni][j] - m_new) * kLog2e);
        }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = (m_run > -INFINITY ? l_run * __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e) : 0.f) + ps;
        m_run = m_new;
    }
    // D = sum_d dO o O over the row's 64 columns: 16 per lane, then across the 4 lanes of the row
    float dsum = 0.f;
    if (qg < p.Tq) {
        const float *orow = p.o + ((int64_t)b * p.Tq + qg) * p.ld_o + h * 64 + 16 * q;
        const float *drow = p.dout + ((int64_t)b * p.Tq + qg) * p.ld_do + h * 64 + 16 * q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a = *reinterpret_cast<const float4 *>(orow + 4 * c), g = *reinterpret_cast<const float4 *>(drow + 4 * c);
            dsum += a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
        }
    }
    dsum += __shfl_xor(dsum, 16);
    dsum += __shfl_xor(dsum, 32);
    if (q == 0 && qg < p.Tq) {
        const int64_t idx = ((int64_t)b * p.H + h) * p.Tq + qg;
        if (!p.have_lse) p.lse[idx] = m_run + __logf(l_run);
        p.dvec[idx] = dsum;
    }
}

// per-row scalars of the two 16-row tiles of this wave's query half; rows beyond Tq get lse = +inf (P = 0)
__device__ __forceinline__ void row_scalars(const BwdParams &p, int b, int h, int i0, int wm, int r, float (&lse)[2], float (&dv)[2]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int qg = i0 + 32 * wm + 16 * mi + r;
        const int64_t idx = ((int64_t)b * p.H + h) * p.Tq + min(qg, p.Tq - 1);
        lse[mi] = qg < p.Tq ? p.lse[idx] : INFINITY;
        dv[mi] = qg < p.Tq ? p.dvec[idx] : 0.f;
    }
}

// ---- dK, dV: one workgroup per key block -----------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void attn_bwd_kv_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Kt = lds, *Vt = lds + TILE, *Qt = lds + 2 * TILE, *dOt = lds + 3 * TILE, *Pt = lds + 4 * TILE, *dSt = lds + 5 * TILE;
    const int j0 = blockIdx.x * BT, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4, wm = w / WN, wn = w % WN;
    const float *qb = p.q + (int64_t)b * p.Tq * p.ld_q + h * 64;
    const float *dob = p.dout + (int64_t)b * p.Tq * p.ld_do + h * 64;
    const float *kb = p.k + (int64_t)b * p.Tk * p.ld_kv + h * 64;
    const float *vb = p.v + (int64_t)b * p.Tk * p.ld_kv + h * 64;
    load_tile<NT>(Kt, kb, p.ld_kv, j0, p.Tk, tid);
    load_tile<NT>(Vt, vb, p.ld_kv, j0, p.Tk, tid);
    f32x4 dK[2][NI], dV[2][NI];
    zero(dK);
    zero(dV);
    const int nqb = (p.Tq + BT - 1) / BT;
    const int ib0 = p.causal ? j0 / BT : 0;                         // causal: queries before this key block see none of its keys
    float4 qn[1024 / NT], don[1024 / NT];                           // the next query block's Q / dO rows, in flight under the products
    fetch_tile<NT>(qn, qb, p.ld_q, ib0 * BT, p.Tq, tid);
    fetch_tile<NT>(don, dob, p.ld_do, ib0 * BT, p.Tq, tid);
    for (int ib = ib0; ib < nqb; ++ib) {
        const int i0 = ib * BT;
        __syncthreads();                                            // the previous iteration's reads of Qt / dOt / Pt / dSt
        put_tile<NT>(Qt, qn, tid);
        put_tile<NT>(dOt, don, tid);
        float lse[2], dv[2];
        row_scalars(p, b, h, i0, wm, r, lse, dv);
        __syncthreads();
        if (ib + 1 < nqb) {
            fetch_tile<NT>(qn, qb, p.ld_q, i0 + BT, p.Tq, tid);
            fetch_tile<NT>(don, dob, p.ld_do, i0 + BT, p.Tq, tid);
        }
        f32x4 P[2][NI], dS[2][NI];
        tile_p_ds(p, Qt, dOt, Kt, Vt, i0, j0, lse, dv, wm, wn, r, q, P, dS);
        store_acc_tile(Pt, P, wm, wn, r, q);
        store_acc_tile(dSt, dS, wm, wn, r, q);
        __syncthreads();
        mma64<true, true>(dV, Pt, dOt, wm, wn, r, q);               // dV[key][d] += sum_q P[q][key] dO[q][d]
        mma64<true, true>(dK, dSt, Qt, wm, wn, r, q);               // dK[key][d] += sum_q dS[q][key] q[q][d]
    }
    store_acc_global(p.dk + (int64_t)b * p.Tk * p.ld_dkv + h * 64, p.ld_dkv, j0, p.Tk, dK, wm, wn, r, q);
    store_acc_global(p.dv + (int64_t)b * p.Tk * p.ld_dkv + h * 64, p.ld_dkv, j0, p.Tk, dV, wm, wn, r, q);
}

// ---- dQ: one workgroup per query block ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void attn_bwd_q_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Qt = lds, *dOt = lds + TILE, *Kt = lds + 2 * TILE, *Vt = lds + 3 * TILE, *dSt = lds + 4 * TILE;
    const int i0 = blockIdx.x * BT, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4, wm = w / WN, wn = w % WN;
    const float *kb = p.k + (int64_t)b * p.Tk * p.ld_kv + h * 64;
    const float *vb = p.v + (int64_t)b * p.Tk * p.ld_kv + h * 64;
    load_tile<NT>(Qt, p.q + (int64_t)b * p.Tq * p.ld_q + h * 64, p.ld_q, i0, p.Tq, tid);
    load_tile<NT>(dOt, p.dout + (int64_t)b * p.Tq * p.ld_do + h * 64, p.ld_do, i0, p.Tq, tid);
    float lse[2], dv[2];
    row_scalars(p, b, h, i0, wm, r, lse, dv);
    f32x4 dQ[2][NI];
    zero(dQ);
    int nkb = (p.Tk + BT - 1) / BT;
    if (p.causal) nkb = min(nkb, (min(p.Tq, i0 + BT) - 1) / BT + 1);
    float4 kn[1024 / NT], vn[1024 / NT];                            // the next key block's K / V rows, in flight under the products
    fetch_tile<NT>(kn, kb, p.ld_kv, 0, p.Tk, tid);
    fetch_tile<NT>(vn, vb, p.ld_kv, 0, p.Tk, tid);
    for (int jb = 0; jb < nkb; ++jb) {
        const int j0 = jb * BT;
        __syncthreads();
        put_tile<NT>(Kt, kn, tid);
        put_tile<NT>(Vt, vn, tid);
        __syncthreads();
        if (jb + 1 < nkb) {
            fetch_tile<NT>(kn, kb, p.ld_kv, j0 + BT, p.Tk, tid);
            fetch_tile<NT>(vn, vb, p.ld_kv, j0 + BT, p.Tk, tid);
        }
        f32x4 P[2][NI], dS[2][NI];
        tile_p_ds(p, Qt, dOt, Kt, Vt, i0, j0, lse, dv, wm, wn, r, q, P, dS);
        store_acc_tile(dSt, dS, wm, wn, r, q);
        __syncthreads();
        mma64<false, true>(dQ, dSt, Kt, wm, wn, r, q);              // dQ[q][d] += sum_key dS[q][key] k[key][d]
    }
    store_acc_global(p.dq + (int64_t)b * p.Tq * p.ld_dq + h * 64, p.ld_dq, i0, p.Tq, dQ, wm, wn, r, q);
}

}  // namespace

extern "C" int la_attention_bwd_workspace_bytes(int32_t batch, int32_t q_len, int32_t n_head, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && q_len > 0 && n_head > 0, "attention_bwd_workspace_bytes: bad arguments");
    *bytes = (size_t)2 * batch * n_head * q_len * sizeof(float);
    return LA_OK;
}

// The statistics launch alone (lse unless handed in, D = sum dO o O per query row): shared with la_attention_bwd_f16x2 (la_attention_f16x2.hip).
extern "C" int la_attention_bwd_stats_f32(const float *q, int64_t ld_q, const float *k, int64_t ld_kv, const float *o, int64_t ld_o, const float *dout,
                                          int64_t ld_do, int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal,
                                          const float *lse_in, float *lse, float *dvec, void *stream_) {
    LA_CHECK_ARG(q && k && o && dout && lse && dvec && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_bwd_stats: bad arguments");
    BwdParams p{q, k, nullptr, o, dout, nullptr, nullptr, nullptr, ld_q, ld_kv, ld_o, ld_do, 0, 0, batch, q_len, kv_len, n_head, causal ? 1 : 0,
                lse_in ? const_cast<float *>(lse_in) : lse, dvec, lse_in ? 1 : 0};
    hipLaunchKernelGGL(attn_stats_kernel, dim3(la::cdiv(q_len, BT), n_head, batch), dim3(256), 2 * TILE * 4, (hipStream_t)stream_, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_attention_bwd_f32(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, const float *o, int64_t ld_o,
                                    const float *dout, int64_t ld_do, float *dq, int64_t ld_dq, float *dk, float *dv, int64_t ld_dkv,
                                    int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, const float *lse_in,
                                    void *workspace, size_t workspace_bytes, void *stream_) {
    if (batch == 0 || q_len == 0 || kv_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && o && dout && dq && dk && dv && workspace, "attention_bwd: null pointer");
    LA_CHECK_ARG(batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_bwd: bad sizes");
    LA_CHECK_ARG(!causal || q_len == kv_len, "attention_bwd: the causal mask needs q_len == kv_len (as la_attention_ex / la_attention_lse_f32)");
    LA_CHECK_ARG(ld_q % 4 == 0 && ld_kv % 4 == 0 && ld_o % 4 == 0 && ld_do % 4 == 0 && ld_dq % 4 == 0 && ld_dkv % 4 == 0 &&
                     ld_q >= 64 * n_head && ld_kv >= 64 * n_head && ld_o >= 64 * n_head && ld_do >= 64 * n_head && ld_dq >= 64 * n_head && ld_dkv >= 64 * n_head,
                 "attention_bwd: row pitches must be multiples of 4 floats and cover 64 x heads columns");
    LA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0,
                 "attention_bwd: operands must be 16-byte aligned");
    size_t need = 0;
    la_attention_bwd_workspace_bytes(batch, q_len, n_head, &need);
    LA_CHECK_ARG(workspace_bytes >= need && (uintptr_t)workspace % 4 == 0, "attention_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    hipStream_t stream = (hipStream_t)stream_;
    float *lse = static_cast<float *>(workspace);
    BwdParams p{q, k, v, o, dout, dq, dk, dv, ld_q, ld_kv, ld_o, ld_do, ld_dq, ld_dkv, batch, q_len, kv_len, n_head, causal ? 1 : 0,
                lse_in ? const_cast<float *>(lse_in) : lse, lse + (size_t)batch * n_head * q_len, lse_in ? 1 : 0};
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_kv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * TILE * 4));
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * TILE * 4));
        attr_once.mark();
    }
    la::TimerScope ts("attention_bwd_f32", stream);
    const dim3 gq(la::cdiv(q_len, BT), n_head, batch), gk(la::cdiv(kv_len, BT), n_head, batch);
    hipLaunchKernelGGL(attn_stats_kernel, gq, dim3(256), 2 * TILE * 4, stream, p);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, gk, dim3(NT), 6 * TILE * 4, stream, p);
    hipLaunchKernelGGL(attn_bwd_q_kernel, gq, dim3(NT), 5 * TILE * 4, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
