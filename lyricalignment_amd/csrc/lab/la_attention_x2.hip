// la_attention_x2.hip -- EXPERIMENT build only (tools/build_variant.sh lab -DLA_EXPERIMENTS; not part of the shipped library).
// Parity-green and as accurate as the float32 kernels, but only 1.07 x (backward) / 1.12 x (forward) faster per layer and no measurable
// gain on the fine-tune step (profiles/r5_kbench_attn_*_x2.txt, r5_ab_finetune_attention_x2.txt): with the products 5 x cheaper the
// sweeps are bound by memory latency at ONE workgroup per CU (the key sweep holds eight two-plane tiles: 147 KB of LDS; forcing two
// workgroups per CU on the query sweep spilled 40 registers and lost 20 %).  profiles/NOTES.md "Attention on the f16x2 scheme" has
// the per-kernel numbers and what a second attempt should change (separate dV / dK sweeps at two workgroups per CU).
//
// The fused float32 attention backward (la_attention_bwd.hip: statistics, key-block sweep for dK / dV,
// query-block sweep for dQ; nothing of size Tq x Tk leaves the CU, no atomics) with its five products on the f16 matrix pipe at
// float32 accuracy (la_f32x2.hip: x s = hi + lo in IEEE half, a b = a_lo b_hi + a_hi b_lo + a_hi b_hi in float32 accumulate).
// After the Linear layers moved to that scheme the float32-MFMA sweeps were 22 % of the fine-tune optimizer step (16x16x4 f32: 1/16 of
// the 16-bit rate; train_multitask.py:325-326 `loss.backward()` through whisper's MultiHeadAttention.qkv_attention).
//
// Operands.  q (pre-scaled), k, v, dO are split once per call into half planes by the kernels of la_f32x2.hip:
//   * row-scaled planes [tokens][2][64 H] + one inverse scale per token row: operands of the products that contract over the head
//     dimension (S = q k^T, dP = dO v^T) -- a scale must be constant along the contraction index;
//   * column-scaled TRANSPOSED planes [64 H][2][tokens] + one inverse scale per column: q^T, dO^T (dK = dS^T q, dV = P^T dO contract
//     over the queries) and k^T (dQ = dS k contracts over the keys).
// P in [0, 1] is split with the fixed scale 2^14.  dS = P o (dP - D) is split with a scale from a BOUND that needs no reduction over
// the tile: |dS_ij| <= |dP_ij| + |D_i| <= 2 |dO_i| max_j |v_j| (D_i is a convex combination of dP_i.), per query row in the
// query sweep (contraction over keys) and the largest of the block's 64 rows in the key sweep (contraction over queries).  Entries
// more than 2^17 below their bound lose relative -- never absolute -- precision (half's subnormal range): below float32 rounding of
// the sums they enter.  Accumulators whose scale is the same for every block are summed raw and scaled once at the end.
//
// Geometry as in la_attention_bwd.hip: 64 x 64 tiles, 8-wave workgroups as 2 (row halves) x 4 (16-column strips), one
// workgroup per CU (the key sweep holds eight tiles of two planes: 147 KB of LDS).  Each 16 x 16 x 64 block product is six
// v_mfma_f32_16x16x32_f16 (96 cycles) instead of sixteen v_mfma_f32_16x16x4_f32 (512).  The operand orientation of a product is
// chosen so that a lane holds four consecutive elements ALONG the next product's contraction index: P^T / dS^T leave the
// accumulators as 8-byte half rows.
#ifndef LA_EXPERIMENTS
#error "lab/la_attention_x2.hip belongs to the experiment build (-DLA_EXPERIMENTS, tools/build_variant.sh)"
#endif
#include <algorithm>

#include "../la_x2.h"

extern "C" int la_split_f16x2(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale, void *stream);
extern "C" int la_split_f16x2_t(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t, void *stream);

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int BT = 64;
constexpr int PB = 144;                  // bytes per tile row of one plane: 64 halves + 16 (rows start 16 B apart modulo 128 B)
constexpr int PLANE = BT * PB;           // 9216
constexpr int TILE2 = 2 * PLANE;         // hi plane, lo plane
constexpr float kLog2e = 1.4426950408889634f;
constexpr int NT = 512;

struct X2Params {
    // row-scaled planes [tokens][2][C] (C = 64 H) and their per-row inverse scales
    const unsigned short *qp, *kp, *vp, *dop;
    const float *rq, *rk, *rv, *rdo;
    // column-scaled transposed planes [C][2][mp] and their per-column inverse scales
    const unsigned short *qt, *dot, *kt;
    const float *cq, *cdo, *ck;
    int64_t mpq, mpk;                    // plane lengths of the transposed operands (query-side, key-side)
    float *dq, *dk, *dv;
    int64_t ld_dq, ld_dkv;
    int B, Tq, Tk, H, causal;
    const float *lse, *dvec, *nrm;       // [B][H][Tq]: log-sum-exp, D = sum dO o O, |dO_i|
    const unsigned *vmax;                // [B][H]: max_j |v_j| as float bits
};

// ---- |dO_i| per (clip, head, query) and max_j |v_j| per (clip, head) -------------------------------------------------------
__global__ __launch_bounds__(256) void attn_norms_kernel(const float *dout, int64_t ld_do, const float *v, int64_t ld_kv, int B, int Tq, int Tk, int H,
                                                         float *nrm, unsigned *vmax) {
    const int row = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3, h = blockIdx.y, b = blockIdx.z;
    float s1 = 0.f, s2 = 0.f;
    if (row < Tq) {
        const float *p = dout + ((int64_t)b * Tq + row) * ld_do + h * 64 + part * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float4 a = *reinterpret_cast<const float4 *>(p + 4 * c); s1 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
    }
    if (row < Tk) {
        const float *p = v + ((int64_t)b * Tk + row) * ld_kv + h * 64 + part * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float4 a = *reinterpret_cast<const float4 *>(p + 4 * c); s2 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    if (part == 0 && row < Tq) nrm[((int64_t)b * H + h) * Tq + row] = sqrtf(s1);
    float m = sqrtf(s2);
    if (!(m == m)) m = __uint_as_float(0x7f800000u);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(vmax + b * H + h, __float_as_uint(m));
}

// ---- tile movers ------------------------------------------------------------------------------------------------------------
// rows row0 .. row0 + 63 (zero beyond `limit`) x 64 halves at column col0 of row-scaled planes [rows][2][C]: 2 x 16 B per thread
__device__ __forceinline__ void fetch_rows(uint4 (&v)[2], const unsigned short *planes, int64_t C, int64_t tok0, int row0, int limit, int col0, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * NT, plane = idx >> 9, row = (idx >> 3) & 63, c = idx & 7;
        v[it] = make_uint4(0u, 0u, 0u, 0u);
        if (row0 + row < limit) v[it] = *reinterpret_cast<const uint4 *>(planes + ((tok0 + row0 + row) * 2 + plane) * C + col0 + c * 8);
    }
}
__device__ __forceinline__ void put_rows(unsigned char *tile, const uint4 (&v)[2], int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * NT, plane = idx >> 9, row = (idx >> 3) & 63, c = idx & 7;
        *reinterpret_cast<uint4 *>(tile + plane * PLANE + row * PB + c * 16) = v[it];
    }
}
// rows d0 .. d0 + 63 x tokens tok0 + t0 .. + 63 (zero beyond `limit`; tok0, t0, limit multiples of 4) of transposed planes
// [C][2][mp]: 4 x 8 B per thread
__device__ __forceinline__ void fetch_cols(uint2 (&v)[4], const unsigned short *planes_t, int64_t mp, int d0, int64_t tok0, int t0, int limit, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * NT, plane = idx >> 10, row = (idx >> 4) & 63, c = idx & 15;
        v[it] = make_uint2(0u, 0u);
        if (t0 + c * 4 < limit) v[it] = *reinterpret_cast<const uint2 *>(planes_t + ((int64_t)(d0 + row) * 2 + plane) * mp + tok0 + t0 + c * 4);
    }
}
__device__ __forceinline__ void put_cols(unsigned char *tile, const uint2 (&v)[4], int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * NT, plane = idx >> 10, row = (idx >> 4) & 63, c = idx & 15;
        *reinterpret_cast<uint2 *>(tile + plane * PLANE + row * PB + c * 8) = v[it];
    }
}

// acc[t] += sum_k A[arow0 + 4 q + t][k] B[brow0 + r16][k], k = 0 .. 63, both tiles as (hi, lo) planes
__device__ __forceinline__ void mma_x2(f32x4 &acc, const unsigned char *A, int arow0, const unsigned char *B, int brow0, int r16, int q) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ao = (arow0 + r16) * PB + ks * 64 + q * 16, bo = (brow0 + r16) * PB + ks * 64 + q * 16;
        const f16x8 ah = *reinterpret_cast<const f16x8 *>(A + ao), al = *reinterpret_cast<const f16x8 *>(A + PLANE + ao);
        const f16x8 bh = *reinterpret_cast<const f16x8 *>(B + bo), bl = *reinterpret_cast<const f16x8 *>(B + PLANE + bo);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
}
// B fragments (rows brow0 + r16 of a tile: both k-steps, both planes) held in registers across many products
struct BFrag { f16x8 h[2], l[2]; };
__device__ __forceinline__ BFrag load_bfrag(const unsigned char *B, int brow0, int r16, int q) {
    BFrag f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int bo = (brow0 + r16) * PB + ks * 64 + q * 16;
        f.h[ks] = *reinterpret_cast<const f16x8 *>(B + bo);
        f.l[ks] = *reinterpret_cast<const f16x8 *>(B + PLANE + bo);
    }
    return f;
}
__device__ __forceinline__ void mma_x2_breg(f32x4 &acc, const unsigned char *A, int arow0, const BFrag &b, int r16, int q) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ao = (arow0 + r16) * PB + ks * 64 + q * 16;
        const f16x8 ah = *reinterpret_cast<const f16x8 *>(A + ao), al = *reinterpret_cast<const f16x8 *>(A + PLANE + ao);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b.h[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.l[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b.h[ks], acc, 0, 0, 0);
    }
}
// four consecutive elements of one tile row as halves: 8 bytes into each plane
__device__ __forceinline__ void put4(unsigned char *tile, int row, int col, const float (&x)[4], float s, int plane_bytes = PLANE) {
    const unsigned p0 = la::x2::pack_hi_lo(x[0] * s), p1 = la::x2::pack_hi_lo(x[1] * s), p2 = la::x2::pack_hi_lo(x[2] * s), p3 = la::x2::pack_hi_lo(x[3] * s);
    *reinterpret_cast<uint2 *>(tile + row * PB + col * 2) = make_uint2((p0 & 0xffffu) | (p1 << 16), (p2 & 0xffffu) | (p3 << 16));
    *reinterpret_cast<uint2 *>(tile + plane_bytes + row * PB + col * 2) = make_uint2((p0 >> 16) | (p1 & 0xffff0000u), (p2 >> 16) | (p3 & 0xffff0000u));
}

// ---- dK, dV: one workgroup per key block, sweeping the query blocks -----------------------------------------------------------
__global__ __launch_bounds__(NT) void attn_bwd_kv_x2_kernel(X2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *Kt = lds, *Vt = lds + TILE2, *Qt = lds + 2 * TILE2, *dOt = lds + 3 * TILE2, *QTt = lds + 4 * TILE2, *dOTt = lds + 5 * TILE2,
                  *PTt = lds + 6 * TILE2, *dSTt = lds + 7 * TILE2;
    float *s_lse = reinterpret_cast<float *>(lds + 8 * TILE2), *s_d = s_lse + 64, *s_rq = s_lse + 128, *s_rdo = s_lse + 192, *s_bnd = s_lse + 256;
    const int j0 = blockIdx.x * BT, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, q = lane >> 4, wm = w >> 2, wn = w & 3;
    const int64_t C = 64 * (int64_t)p.H;
    const int64_t qtok0 = (int64_t)b * p.Tq, ktok0 = (int64_t)b * p.Tk;
    {
        uint4 t[2];
        fetch_rows(t, p.kp, C, ktok0, j0, p.Tk, h * 64, tid); put_rows(Kt, t, tid);
        fetch_rows(t, p.vp, C, ktok0, j0, p.Tk, h * 64, tid); put_rows(Vt, t, tid);
    }
    __syncthreads();
    const BFrag kf = load_bfrag(Kt, 16 * wn, r16, q), vf = load_bfrag(Vt, 16 * wn, r16, q);      // this wave's key rows: loop invariant
    const int key_l = 16 * wn + r16, kg = j0 + key_l;                       // this lane's key in the S / dP products
    const float rk_n = kg < p.Tk ? p.rk[ktok0 + kg] : 0.f, rv_n = kg < p.Tk ? p.rv[ktok0 + kg] : 0.f;
    const float vmx = __uint_as_float(p.vmax[b * p.H + h]);
    f32x4 dV[2], dK[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) { dV[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; dK[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int nqb = (p.Tq + BT - 1) / BT;
    const int ib0 = p.causal ? j0 / BT : 0;
    uint4 qn[2], don[2];
    uint2 qtn[4], dotn[4];
    fetch_rows(qn, p.qp, C, qtok0, ib0 * BT, p.Tq, h * 64, tid);
    fetch_rows(don, p.dop, C, qtok0, ib0 * BT, p.Tq, h * 64, tid);
    fetch_cols(qtn, p.qt, p.mpq, h * 64, qtok0, ib0 * BT, p.Tq, tid);
    fetch_cols(dotn, p.dot, p.mpq, h * 64, qtok0, ib0 * BT, p.Tq, tid);
    // per-query scalars of the next block (threads 0 .. 63), fetched with its tiles: loaded inside the block they cost it a memory round trip
    float n_lse = INFINITY, n_d = 0.f, n_rq = 0.f, n_rdo = 0.f, n_bnd = 0.f;
    auto fetch_scalars = [&](int i0) {
        if (tid < 64) {
            const int qg = i0 + tid;
            const int64_t idx = ((int64_t)b * p.H + h) * p.Tq + min(qg, p.Tq - 1);
            const bool in = qg < p.Tq;
            n_lse = in ? p.lse[idx] : INFINITY;
            n_d = in ? p.dvec[idx] : 0.f;
            n_rq = in ? p.rq[qtok0 + qg] : 0.f;
            n_rdo = in ? p.rdo[qtok0 + qg] : 0.f;
            n_bnd = in ? 2.0f * p.nrm[idx] * vmx : 0.f;
        }
    };
    fetch_scalars(ib0 * BT);
    for (int ib = ib0; ib < nqb; ++ib) {
        const int i0 = ib * BT;
        __syncthreads();                                                    // the previous block's reads of the query-side tiles / arrays
        put_rows(Qt, qn, tid); put_rows(dOt, don, tid); put_cols(QTt, qtn, tid); put_cols(dOTt, dotn, tid);
        if (tid < 64) { s_lse[tid] = n_lse; s_d[tid] = n_d; s_rq[tid] = n_rq; s_rdo[tid] = n_rdo; s_bnd[tid] = n_bnd; }
        __syncthreads();
        if (ib + 1 < nqb) {
            fetch_rows(qn, p.qp, C, qtok0, i0 + BT, p.Tq, h * 64, tid);
            fetch_rows(don, p.dop, C, qtok0, i0 + BT, p.Tq, h * 64, tid);
            fetch_cols(qtn, p.qt, p.mpq, h * 64, qtok0, i0 + BT, p.Tq, tid);
            fetch_cols(dotn, p.dot, p.mpq, h * 64, qtok0, i0 + BT, p.Tq, tid);
            fetch_scalars(i0 + BT);
        }
        // scale of this block's dS^T tile from the largest row bound (no reduction over the tile's values)
        float bt = s_bnd[lane];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) bt = fmaxf(bt, __shfl_xor(bt, o));
        float inv_st;
        const float st = la::x2::scale_for(bt, &inv_st);
        // S[q][key] and dP[q][key]: lane = four consecutive queries (rows of the A operand) of one key
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int q0 = 32 * wm + 16 * mi;
            f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_x2_breg(S, Qt, q0, kf, r16, q);
            mma_x2_breg(dP, dOt, q0, vf, r16, q);
            float pv[4], ds[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ql = q0 + 4 * q + t, qg = i0 + ql;
                const bool ok = kg < p.Tk && qg < p.Tq && (!p.causal || kg <= qg);
                const float s = S[t] * (s_rq[ql] * rk_n), d = dP[t] * (s_rdo[ql] * rv_n);
                pv[t] = ok ? __builtin_amdgcn_exp2f((s - s_lse[ql]) * kLog2e) : 0.f;
                ds[t] = pv[t] * (d - s_d[ql]);
            }
            put4(PTt, key_l, q0 + 4 * q, pv, 16384.0f);
            put4(dSTt, key_l, q0 + 4 * q, ds, st);
        }
        __syncthreads();
        // dV[d][key] += sum_q dO^T[d][q] P^T[key][q];  dK[d][key] += (sum_q q^T[d][q] dS^T[key][q]) / st
        {
            const BFrag pf = load_bfrag(PTt, 16 * wn, r16, q), sf = load_bfrag(dSTt, 16 * wn, r16, q);   // one read for both row halves
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int d0 = 32 * wm + 16 * mi;
                mma_x2_breg(dV[mi], dOTt, d0, pf, r16, q);
                f32x4 kb = f32x4{0.f, 0.f, 0.f, 0.f};
                mma_x2_breg(kb, QTt, d0, sf, r16, q);
#pragma unroll
                for (int t = 0; t < 4; ++t) dK[mi][t] = fmaf(kb[t], inv_st, dK[mi][t]);
            }
        }
    }
    if (kg < p.Tk) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int d = h * 64 + 32 * wm + 16 * mi + 4 * q;
            const float4 cd = *reinterpret_cast<const float4 *>(p.cdo + d), cq4 = *reinterpret_cast<const float4 *>(p.cq + d);
            constexpr float ip = 6.103515625e-05f;                            // 2^-14: P's scale
            *reinterpret_cast<float4 *>(p.dv + (ktok0 + kg) * p.ld_dkv + d) = make_float4(dV[mi][0] * (ip * cd.x), dV[mi][1] * (ip * cd.y), dV[mi][2] * (ip * cd.z), dV[mi][3] * (ip * cd.w));
            *reinterpret_cast<float4 *>(p.dk + (ktok0 + kg) * p.ld_dkv + d) = make_float4(dK[mi][0] * cq4.x, dK[mi][1] * cq4.y, dK[mi][2] * cq4.z, dK[mi][3] * cq4.w);
        }
    }
}

// ---- dQ: one workgroup per query block, sweeping the key blocks -----------------------------------------------------------------
__global__ __launch_bounds__(NT) void attn_bwd_q_x2_kernel(X2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // Q / dO are only staged to be read back as this wave's loop-invariant B fragments: their tiles share the space of K / V (74 KB per
    // workgroup: two workgroups per CU, so one's memory round trips hide under the other's products)
    unsigned char *Qt = lds, *dOt = lds + TILE2, *Kt = lds, *Vt = lds + TILE2, *KTt = lds + 2 * TILE2, *dSt = lds + 3 * TILE2;
    float *s_rk = reinterpret_cast<float *>(lds + 4 * TILE2), *s_rv = s_rk + 64;
    const int i0 = blockIdx.x * BT, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, q = lane >> 4, wm = w >> 2, wn = w & 3;
    const int64_t C = 64 * (int64_t)p.H;
    const int64_t qtok0 = (int64_t)b * p.Tq, ktok0 = (int64_t)b * p.Tk;
    {
        uint4 t[2];
        fetch_rows(t, p.qp, C, qtok0, i0, p.Tq, h * 64, tid); put_rows(Qt, t, tid);
        fetch_rows(t, p.dop, C, qtok0, i0, p.Tq, h * 64, tid); put_rows(dOt, t, tid);
    }
    __syncthreads();
    const BFrag qf = load_bfrag(Qt, 16 * wn, r16, q), dof = load_bfrag(dOt, 16 * wn, r16, q);    // this wave's query rows: loop invariant
    // this lane's query (the B operand's row in S^T / dP^T, the column of dQ^T): loop invariant
    const int q_l = 16 * wn + r16, qg = i0 + q_l;
    const bool q_in = qg < p.Tq;
    const int64_t sidx = ((int64_t)b * p.H + h) * p.Tq + min(qg, p.Tq - 1);
    const float lse_n = q_in ? p.lse[sidx] : INFINITY, d_n = q_in ? p.dvec[sidx] : 0.f;
    const float rq_n = q_in ? p.rq[qtok0 + qg] : 0.f, rdo_n = q_in ? p.rdo[qtok0 + qg] : 0.f;
    float inv_sq;
    const float sq = la::x2::scale_for(q_in ? 2.0f * p.nrm[sidx] * __uint_as_float(p.vmax[b * p.H + h]) : 0.f, &inv_sq);
    f32x4 dQ[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) dQ[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    int nkb = (p.Tk + BT - 1) / BT;
    if (p.causal) nkb = min(nkb, (min(p.Tq, i0 + BT) - 1) / BT + 1);
    uint4 kn[2], vn[2];
    uint2 ktn[4];
    fetch_rows(kn, p.kp, C, ktok0, 0, p.Tk, h * 64, tid);
    fetch_rows(vn, p.vp, C, ktok0, 0, p.Tk, h * 64, tid);
    fetch_cols(ktn, p.kt, p.mpk, h * 64, ktok0, 0, p.Tk, tid);
    float n_rk = 0.f, n_rv = 0.f;                                         // the next block's key scales (threads 0 .. 63), fetched with its tiles
    auto fetch_scalars = [&](int j0) {
        if (tid < 64) {
            const int kg = j0 + tid;
            n_rk = kg < p.Tk ? p.rk[ktok0 + kg] : 0.f;
            n_rv = kg < p.Tk ? p.rv[ktok0 + kg] : 0.f;
        }
    };
    fetch_scalars(0);
    for (int jb = 0; jb < nkb; ++jb) {
        const int j0 = jb * BT;
        __syncthreads();
        put_rows(Kt, kn, tid); put_rows(Vt, vn, tid); put_cols(KTt, ktn, tid);
        if (tid < 64) { s_rk[tid] = n_rk; s_rv[tid] = n_rv; }
        __syncthreads();
        if (jb + 1 < nkb) {
            fetch_rows(kn, p.kp, C, ktok0, j0 + BT, p.Tk, h * 64, tid);
            fetch_rows(vn, p.vp, C, ktok0, j0 + BT, p.Tk, h * 64, tid);
            fetch_cols(ktn, p.kt, p.mpk, h * 64, ktok0, j0 + BT, p.Tk, tid);
            fetch_scalars(j0 + BT);
        }
        // S^T[key][q], dP^T[key][q]: lane = four consecutive keys (rows of the A operand) of one query
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int k0 = 32 * wm + 16 * mi;
            f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_x2_breg(S, Kt, k0, qf, r16, q);
            mma_x2_breg(dP, Vt, k0, dof, r16, q);
            float ds[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kl = k0 + 4 * q + t, kg = j0 + kl;
                const bool ok = kg < p.Tk && q_in && (!p.causal || kg <= qg);
                const float s = S[t] * (s_rk[kl] * rq_n), d = dP[t] * (s_rv[kl] * rdo_n);
                const float pv = ok ? __builtin_amdgcn_exp2f((s - lse_n) * kLog2e) : 0.f;
                ds[t] = pv * (d - d_n);
            }
            put4(dSt, q_l, k0 + 4 * q, ds, sq);
        }
        __syncthreads();
        // dQ[d][q] += sum_key k^T[d][key] dS[q][key]   (one scale per query for every block: summed raw)
        {
            const BFrag sf = load_bfrag(dSt, 16 * wn, r16, q);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) mma_x2_breg(dQ[mi], KTt, 32 * wm + 16 * mi, sf, r16, q);
        }
    }
    if (q_in) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int d = h * 64 + 32 * wm + 16 * mi + 4 * q;
            const float4 c4 = *reinterpret_cast<const float4 *>(p.ck + d);
            *reinterpret_cast<float4 *>(p.dq + (qtok0 + qg) * p.ld_dq + d) =
                make_float4(dQ[mi][0] * (c4.x * inv_sq), dQ[mi][1] * (c4.y * inv_sq), dQ[mi][2] * (c4.z * inv_sq), dQ[mi][3] * (c4.w * inv_sq));
        }
    }
}

// ---- forward: O = softmax(q k^T) v and lse, one workgroup (4 waves) per 64 queries, 16 queries per wave ---------------------------
// The float32 training forward (la_attention_lse_f32: v_mfma_f32_32x32x2f32) on the same scheme.  A wave owns 16 query rows for the
// whole key sweep: S^T[key][q] = k q^T with the K block's rows as the A operand and the wave's q rows (registers) as B, so a lane holds
// 16 keys of ONE query and the online softmax is lane-local + two shuffles; P (scale 2^14, relative to the running maximum) goes
// through the wave's own LDS strip as half planes and comes back as the B operand of O^T[d][q] = v^T P^T (A = the V block's
// column-scaled transposed planes).  The running sums are rescaled per query in registers.
struct X2FwdParams {
    const unsigned short *qp, *kp, *vt;      // q, k: row-scaled planes [tokens][2][C]; v: column-scaled transposed planes [C][2][mp]
    const float *rq, *rk, *cv;
    int64_t mpk;
    float *out, *lse;
    int64_t ld_out;
    int B, Tq, Tk, H, causal;
};

__global__ __launch_bounds__(256, 2) void attn_fwd_x2_kernel(X2FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *Kt = lds, *VTt = lds + TILE2, *Pw = lds + 2 * TILE2;       // Pw: 4 waves x (2 planes x 16 rows x PB)
    float *s_rk = reinterpret_cast<float *>(lds + 2 * TILE2 + 4 * 2 * 16 * PB);
    const int i0 = blockIdx.x * BT, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const int64_t C = 64 * (int64_t)p.H;
    const int64_t qtok0 = (int64_t)b * p.Tq, ktok0 = (int64_t)b * p.Tk;
    unsigned char *Pt = Pw + w * (2 * 16 * PB);                                // hi plane at Pt, lo plane at Pt + 16 * PB
    const int qg = i0 + 16 * w + r16;
    const bool q_in = qg < p.Tq;
    const int qrow = q_in ? qg : p.Tq - 1;
    // this wave's q rows as B fragments: row r16, k = 32 ks + 8 q .. + 7 of each plane
    f16x8 qh[2], ql[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned short *src = p.qp + ((qtok0 + qrow) * 2) * C + h * 64 + ks * 32 + q * 8;
        qh[ks] = *reinterpret_cast<const f16x8 *>(src);
        ql[ks] = *reinterpret_cast<const f16x8 *>(src + C);
    }
    const float rq_n = p.rq[qtok0 + qrow];
    f32x4 o[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) o[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;                                      // exp2 domain; l: this lane's share of the row sum
    int nkb = (p.Tk + BT - 1) / BT;
    if (p.causal) nkb = min(nkb, (min(p.Tq, i0 + BT) - 1) / BT + 1);
    // the next block's K rows (4 x 16 B per thread), V^T (8 x 8 B) and key scales are fetched into registers under the current block's products
    uint4 kn[4];
    uint2 vn[8];
    float rkn = 0.f;
    auto fetch = [&](int j0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256, plane = idx >> 9, row = (idx >> 3) & 63, c = idx & 7;
            kn[it] = make_uint4(0u, 0u, 0u, 0u);
            if (j0 + row < p.Tk) kn[it] = *reinterpret_cast<const uint4 *>(p.kp + ((ktok0 + j0 + row) * 2 + plane) * C + h * 64 + c * 8);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256, plane = idx >> 10, row = (idx >> 4) & 63, c = idx & 15;
            vn[it] = make_uint2(0u, 0u);
            if (j0 + c * 4 < p.Tk) vn[it] = *reinterpret_cast<const uint2 *>(p.vt + ((int64_t)(h * 64 + row) * 2 + plane) * p.mpk + ktok0 + j0 + c * 4);
        }
        if (tid < 64) rkn = j0 + tid < p.Tk ? p.rk[ktok0 + j0 + tid] : 0.f;
    };
    fetch(0);
    for (int jb = 0; jb < nkb; ++jb) {
        const int j0 = jb * BT;
        __syncthreads();                                                       // the previous block's reads of Kt / VTt
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256, plane = idx >> 9, row = (idx >> 3) & 63, c = idx & 7;
            *reinterpret_cast<uint4 *>(Kt + plane * PLANE + row * PB + c * 16) = kn[it];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256, plane = idx >> 10, row = (idx >> 4) & 63, c = idx & 15;
            *reinterpret_cast<uint2 *>(VTt + plane * PLANE + row * PB + c * 8) = vn[it];
        }
        if (tid < 64) s_rk[tid] = rkn;
        __syncthreads();
        if (jb + 1 < nkb) fetch(j0 + BT);
        // S^T[key][q]: four 16-key tiles; lane = keys 16 mi + 4 q + t of query r16
        float sv[4][4];
        float mx = -INFINITY;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ao = (16 * mi + r16) * PB + ks * 64 + q * 16;
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(Kt + ao), al = *reinterpret_cast<const f16x8 *>(Kt + PLANE + ao);
                S = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh[ks], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql[ks], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh[ks], S, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kl = 16 * mi + 4 * q + t, kg = j0 + kl;
                const bool ok = kg < p.Tk && (!p.causal || kg <= qg);
                sv[mi][t] = ok ? S[t] * (s_rk[kl] * rq_n) * kLog2e : -INFINITY;
                mx = fmaxf(mx, sv[mi][t]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = m_new > -INFINITY ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.f;      // (m_run = -inf: exp2(-inf) = 0)
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float pv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pv[t] = m_new > -INFINITY ? __builtin_amdgcn_exp2f(sv[mi][t] - m_new) : 0.f;
                psum += pv[t];
            }
            put4(Pt, r16, 16 * mi + 4 * q, pv, 16384.0f, 16 * PB);
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int t = 0; t < 4; ++t) o[mi][t] *= alpha;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // the wave's own P strip: written above, read below
        // O^T[d][q] += sum_key v^T[d][key] P[q][key]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int bo = r16 * PB + ks * 64 + q * 16;
            const f16x8 bh = *reinterpret_cast<const f16x8 *>(Pt + bo), bl = *reinterpret_cast<const f16x8 *>(Pt + 16 * PB + bo);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int ao = (16 * mi + r16) * PB + ks * 64 + q * 16;
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(VTt + ao), al = *reinterpret_cast<const f16x8 *>(VTt + PLANE + ao);
                o[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, o[mi], 0, 0, 0);
                o[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, o[mi], 0, 0, 0);
                o[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, o[mi], 0, 0, 0);
            }
        }
    }
    float l = l_run + __shfl_xor(l_run, 16);
    l += __shfl_xor(l, 32);
    if (q_in) {
        const float inv = 6.103515625e-05f / l;                                // 2^-14 (P's scale) / row sum
        float *orow = p.out + (qtok0 + qg) * p.ld_out + h * 64;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int d = 16 * mi + 4 * q;
            const float4 c4 = *reinterpret_cast<const float4 *>(p.cv + h * 64 + d);
            *reinterpret_cast<float4 *>(orow + d) = make_float4(o[mi][0] * (c4.x * inv), o[mi][1] * (c4.y * inv), o[mi][2] * (c4.z * inv), o[mi][3] * (c4.w * inv));
        }
        if (q == 0) p.lse[((int64_t)b * p.H + h) * p.Tq + qg] = fmaf(m_run, 0.6931471805599453f, __logf(l));
    }
}

struct X2Plan {
    size_t qp, kp, vp, dop, qt, dot, kt, rq, rk, rv, rdo, cq, cdo, ck, nrm, vmax, stats, total;
    int64_t mpq, mpk;
};
X2Plan plan_x2(int B, int Tq, int Tk, int H) {
    X2Plan pl;
    const size_t C = (size_t)64 * H, Mq = (size_t)B * Tq, Mk = (size_t)B * Tk;
    pl.mpq = la::round_up((int64_t)Mq, 64);
    pl.mpk = la::round_up((int64_t)Mk, 64);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += la::round_up((int64_t)bytes, 256); return o; };
    pl.qp = take(Mq * 2 * C * 2); pl.dop = take(Mq * 2 * C * 2); pl.kp = take(Mk * 2 * C * 2); pl.vp = take(Mk * 2 * C * 2);
    pl.qt = take(C * 2 * pl.mpq * 2); pl.dot = take(C * 2 * pl.mpq * 2); pl.kt = take(C * 2 * pl.mpk * 2);
    pl.rq = take(Mq * 4); pl.rdo = take(Mq * 4); pl.rk = take(Mk * 4); pl.rv = take(Mk * 4);
    pl.cq = take(C * 4); pl.cdo = take(C * 4); pl.ck = take(C * 4);
    pl.nrm = take((size_t)B * H * Tq * 4); pl.vmax = take((size_t)B * H * 4);
    pl.stats = take((size_t)2 * B * H * Tq * 4);                          // lse + D of la_attention_bwd_f32's statistics launch
    pl.total = off;
    return pl;
}

}  // namespace

extern "C" int la_attention_bwd_workspace_bytes(int32_t batch, int32_t q_len, int32_t n_head, size_t *bytes);
extern "C" int la_attention_bwd_stats_f32(const float *q, int64_t ld_q, const float *k, int64_t ld_kv, const float *o, int64_t ld_o, const float *dout,
                                          int64_t ld_do, int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal,
                                          const float *lse_in, float *lse, float *dvec, void *stream);

extern "C" int la_attention_bwd_x2_workspace_bytes(int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_bwd_x2_workspace_bytes: bad arguments");
    *bytes = plan_x2(batch, q_len, kv_len, n_head).total;
    return LA_OK;
}

// Same contract as la_attention_bwd_f32 (q pre-scaled; dq the gradient with respect to that q; lse_in optional), q_len and kv_len
// multiples of 4 (LA_EUNSUPPORTED otherwise: the caller keeps la_attention_bwd_f32).
extern "C" int la_attention_bwd_x2_f32(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, const float *o, int64_t ld_o,
                                       const float *dout, int64_t ld_do, float *dq, int64_t ld_dq, float *dk, float *dv, int64_t ld_dkv,
                                       int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, const float *lse_in,
                                       void *workspace, size_t workspace_bytes, void *stream_) {
    if (batch == 0 || q_len == 0 || kv_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && o && dout && dq && dk && dv && workspace, "attention_bwd_x2: null pointer");
    LA_CHECK_ARG(batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_bwd_x2: bad sizes");
    LA_CHECK_ARG(!causal || q_len == kv_len, "attention_bwd_x2: the causal mask needs q_len == kv_len");
    if (q_len % 4 != 0 || kv_len % 4 != 0) {
        la::set_error("attention_bwd_x2: q_len = %d / kv_len = %d must be multiples of 4 (use la_attention_bwd_f32)", q_len, kv_len);
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(ld_q % 4 == 0 && ld_kv % 4 == 0 && ld_o % 4 == 0 && ld_do % 4 == 0 && ld_dq % 4 == 0 && ld_dkv % 4 == 0 &&
                     ld_q >= 64 * n_head && ld_kv >= 64 * n_head && ld_o >= 64 * n_head && ld_do >= 64 * n_head && ld_dq >= 64 * n_head && ld_dkv >= 64 * n_head,
                 "attention_bwd_x2: row pitches must be multiples of 4 floats and cover 64 x heads columns");
    LA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0,
                 "attention_bwd_x2: operands must be 16-byte aligned");
    const X2Plan pl = plan_x2(batch, q_len, kv_len, n_head);
    LA_CHECK_ARG(workspace_bytes >= pl.total && (uintptr_t)workspace % 256 == 0, "attention_bwd_x2: workspace too small (%zu < %zu) or not 256-byte aligned",
                 workspace_bytes, pl.total);
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    auto F = [&](size_t o) { return reinterpret_cast<float *>(ws + o); };
    auto Hp = [&](size_t o) { return reinterpret_cast<unsigned short *>(ws + o); };
    const int C = 64 * n_head, Mq = batch * q_len, Mk = batch * kv_len;
    float *lse = F(pl.stats), *dvec = lse + (size_t)batch * n_head * q_len;
    int rc = la_attention_bwd_stats_f32(q, ld_q, k, ld_kv, o, ld_o, dout, ld_do, batch, q_len, kv_len, n_head, causal, lse_in, lse, dvec, stream_);
    if (rc != LA_OK) return rc;
    LA_HIP(hipMemsetAsync(ws + pl.vmax, 0, (size_t)batch * n_head * 4, stream));
    hipLaunchKernelGGL(attn_norms_kernel, dim3(la::cdiv(std::max(q_len, kv_len), 64), n_head, batch), dim3(256), 0, stream, dout, ld_do, v, ld_kv, batch,
                       q_len, kv_len, n_head, F(pl.nrm), reinterpret_cast<unsigned *>(ws + pl.vmax));
    LA_LAUNCH_CHECK();
    if ((rc = la_split_f16x2(q, ld_q, Mq, C, Hp(pl.qp), C, F(pl.rq), stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2(dout, ld_do, Mq, C, Hp(pl.dop), C, F(pl.rdo), stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2(k, ld_kv, Mk, C, Hp(pl.kp), C, F(pl.rk), stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2(v, ld_kv, Mk, C, Hp(pl.vp), C, F(pl.rv), stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2_t(q, ld_q, Mq, C, Hp(pl.qt), pl.mpq, F(pl.cq), stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2_t(dout, ld_do, Mq, C, Hp(pl.dot), pl.mpq, F(pl.cdo), stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2_t(k, ld_kv, Mk, C, Hp(pl.kt), pl.mpk, F(pl.ck), stream_)) != LA_OK) return rc;
    X2Params p{Hp(pl.qp), Hp(pl.kp), Hp(pl.vp), Hp(pl.dop), F(pl.rq), F(pl.rk), F(pl.rv), F(pl.rdo), Hp(pl.qt), Hp(pl.dot), Hp(pl.kt),
               F(pl.cq), F(pl.cdo), F(pl.ck), pl.mpq, pl.mpk, dq, dk, dv, ld_dq, ld_dkv, batch, q_len, kv_len, n_head, causal ? 1 : 0,
               lse_in ? lse_in : lse, dvec, F(pl.nrm), reinterpret_cast<const unsigned *>(ws + pl.vmax)};
    constexpr int LDS_KV = 8 * TILE2 + 5 * 64 * 4, LDS_Q = 4 * TILE2 + 2 * 64 * 4;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_kv_x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV));
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_q_x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_Q));
        attr_once.mark();
    }
    la::TimerScope ts("attention_bwd_f32", stream);
    hipLaunchKernelGGL(attn_bwd_kv_x2_kernel, dim3(la::cdiv(kv_len, BT), n_head, batch), dim3(NT), LDS_KV, stream, p);
    hipLaunchKernelGGL(attn_bwd_q_x2_kernel, dim3(la::cdiv(q_len, BT), n_head, batch), dim3(NT), LDS_Q, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_attention_x2_workspace_bytes(int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_x2_workspace_bytes: bad arguments");
    const size_t C = (size_t)64 * n_head, Mq = (size_t)batch * q_len, Mk = (size_t)batch * kv_len, mpk = la::round_up((int64_t)Mk, 64);
    *bytes = la::round_up((int64_t)(Mq * 2 * C * 2), 256) + la::round_up((int64_t)(Mk * 2 * C * 2), 256) + la::round_up((int64_t)(C * 2 * mpk * 2), 256) +
             la::round_up((int64_t)(Mq * 4), 256) + la::round_up((int64_t)(Mk * 4), 256) + la::round_up((int64_t)(C * 4), 256);
    return LA_OK;
}

// la_attention_lse_f32 (the float32 training forward: out and lse) with its two products on the f16 pipe at float32 accuracy.
// q_len and kv_len multiples of 4 (else LA_EUNSUPPORTED: keep la_attention_lse_f32).
extern "C" int la_attention_x2_lse_f32(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, float *out, int64_t ld_out,
                                       int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, float *lse, void *workspace,
                                       size_t workspace_bytes, void *stream_) {
    if (batch == 0 || q_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && out && lse && workspace && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_x2_lse: bad arguments");
    LA_CHECK_ARG(!causal || q_len == kv_len, "attention_x2_lse: causal masking is defined for self-attention (q_len == kv_len)");
    if (q_len % 4 != 0 || kv_len % 4 != 0) {
        la::set_error("attention_x2_lse: q_len = %d / kv_len = %d must be multiples of 4 (use la_attention_lse_f32)", q_len, kv_len);
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(ld_q >= n_head * 64 && ld_kv >= n_head * 64 && ld_out >= n_head * 64 && ld_q % 4 == 0 && ld_kv % 4 == 0 && ld_out % 4 == 0 &&
                     ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "attention_x2_lse: row pitches / alignment");
    size_t need = 0;
    la_attention_x2_workspace_bytes(batch, q_len, kv_len, n_head, &need);
    LA_CHECK_ARG(workspace_bytes >= need && (uintptr_t)workspace % 256 == 0, "attention_x2_lse: workspace too small (%zu < %zu) or not 256-byte aligned",
                 workspace_bytes, need);
    hipStream_t stream = (hipStream_t)stream_;
    const size_t C = (size_t)64 * n_head, Mq = (size_t)batch * q_len, Mk = (size_t)batch * kv_len;
    const int64_t mpk = la::round_up((int64_t)Mk, 64);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *o = ws + off; off += la::round_up((int64_t)bytes, 256); return o; };
    unsigned short *qp = reinterpret_cast<unsigned short *>(take(Mq * 2 * C * 2)), *kp = reinterpret_cast<unsigned short *>(take(Mk * 2 * C * 2)),
                   *vt = reinterpret_cast<unsigned short *>(take(C * 2 * mpk * 2));
    float *rq = reinterpret_cast<float *>(take(Mq * 4)), *rk = reinterpret_cast<float *>(take(Mk * 4)), *cv = reinterpret_cast<float *>(take(C * 4));
    int rc;
    if ((rc = la_split_f16x2(q, ld_q, (int)Mq, (int)C, qp, (int64_t)C, rq, stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2(k, ld_kv, (int)Mk, (int)C, kp, (int64_t)C, rk, stream_)) != LA_OK) return rc;
    if ((rc = la_split_f16x2_t(v, ld_kv, (int)Mk, (int)C, vt, mpk, cv, stream_)) != LA_OK) return rc;
    X2FwdParams p{qp, kp, vt, rq, rk, cv, mpk, out, lse, ld_out, batch, q_len, kv_len, n_head, causal ? 1 : 0};
    constexpr int LDS_F = 2 * TILE2 + 4 * 2 * 16 * PB + 64 * 4;
    la::TimerScope ts("attention_f32", stream);
    hipLaunchKernelGGL(attn_fwd_x2_kernel, dim3(la::cdiv(q_len, BT), n_head, batch), dim3(256), LDS_F, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
