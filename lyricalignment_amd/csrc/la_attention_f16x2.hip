// la_attention_f16x2.hip -- the float32 attention of the fine-tune step on the f16 matrix pipe at float32 accuracy ("f16x2", la_f32x2.hip /
// la_x2.h: a float32 value scaled by a power of two is hi + lo in IEEE half, a product a b = a_lo b_hi + a_hi b_lo + a_hi b_hi in float32
// accumulate).  The reference trains in float32 (train_multitask.py:325-326 through whisper/model.py MultiHeadAttention.qkv_attention);
// gfx950 multiplies float32 operands at 1/16 of its 16-bit rate, and after the Linear layers moved to the f16x2 GEMM the float32-MFMA
// attention kernels (la_attention.hip attention_f32_kernel, la_attention_bwd.hip) were a third of the optimizer step.
//
// Forward (this file's first half): the register-resident flash form of the 16-bit inference kernel (la_attention.hip: the score tile is
// computed transposed, S^T = K Q^T, so a lane holds one query's scores; the exponentiated tile is directly the B operand of
// O^T += V^T P^T; K and V tiles staged by LDS-DMA into swizzled images, V^T fragments by ds_read_b64_tr_b16) with every product as three
// 32x32x16 f16 MFMAs.  An earlier attempt built on the 64 x 64 LDS-tiled float32 kernels (round 5, profiles/NOTES.md) was bound by LDS round
// trips at one workgroup per CU and gained 1.1 x; this form keeps Q, the scores, P and O in registers.
//
// Operands.  q (pre-scaled by the caller as for the float32 kernel), k, v are split once per call into planes [token][2][64 H] f16 with ONE
// power-of-two scale per (clip, head) and operand -- the largest magnitude of that head's [T][64] slice lands in [2^13, 2^14).  A common
// scale is what lets the tiles go through the matrix pipe as they lie (no per-row factors inside the tile loop); elements more than 2^17
// below their slice's maximum lose relative (never absolute: 2^-38 of the maximum) precision, far below the float32 rounding of the
// 64-term sums they enter.  P = exp2(s - m) in [0, 1] is split with the fixed scale 2^13, folded into the exponent (exp2(s - m + 13)): the
// row sums carry the same factor and it cancels in O = sum P v / sum P.
#include <type_traits>

#include "la_x2.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#ifndef LA_X2F_KO
#define LA_X2F_KO 0                     // timing-only builds (tools/ab_x2f_knockouts.sh; results are garbage): bit mask of parts of the forward
#endif                                  // tile loop left out -- 1 exponentials, 2 V^T fragment reads, 4 K fragment reads, 8 staging, 16 MFMAs, 32 barrier
constexpr int KT = 64;                  // keys per tile
constexpr int IMG = KT * 128;           // one plane of one K or V tile: [64 keys][128 B]
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct FwdParams {
    const unsigned short *qp, *kp, *vp;      // planes [token][2][C], C = 64 H
    const float *sq, *sk, *sv;               // inverse scales [batch][n_head]
    float *out;
    int64_t ld_out;
    float *lse;                              // [batch][n_head][q_len]
    int q_len, kv_len, n_head, causal, batch;
};

// ---- operand split: one workgroup per (head, clip) ------------------------------------------------------------------------------------
// x rows [clip * T + t][head * 64 ..] (row pitch ld) -> planes [(clip * T + t) * 2 + plane][C] and inv_scale[clip * H + head]
// gridDim.z = parts: with few (clip, head) slices (a micro-batch of two clips: 32) every slice is taken by `parts` workgroups -- each finds the
// slice's maximum itself (the slice is L2-resident) and writes every parts-th group of 64 token rows
__global__ __launch_bounds__(1024) void heads_split_kernel(const float *x, int64_t ld, int T, int H, unsigned short *planes, float *inv_scale) {
    __shared__ float red[16];
    const int head = blockIdx.x, clip = blockIdx.y, tid = threadIdx.x;
    const int c4 = tid & 15, r0 = tid >> 4;
    const float *src = x + (int64_t)clip * T * ld + head * 64 + c4 * 4;
    // slices of up to 1536 tokens (a 30 s clip: 1500) stay in registers between the two passes: 24 float4 per thread, one read of x
    constexpr int NV = 24;
    const bool cached = T <= 64 * NV;
    float4 v[NV];
    float mx = 0.f;
    if (cached) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int r = r0 + 64 * j;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < T) v[j] = *reinterpret_cast<const float4 *>(src + (int64_t)r * ld);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
    } else {
        for (int r = r0; r < T; r += 64) {
            const float4 w = *reinterpret_cast<const float4 *>(src + (int64_t)r * ld);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))));
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
    float inv;
    const float s = la::x2::scale_for(mx, &inv);
    if (tid == 0 && blockIdx.z == 0) inv_scale[clip * H + head] = inv;
    const int64_t C = 64 * (int64_t)H;
    const int parts = gridDim.z, part = blockIdx.z;
    unsigned short *dst = planes + (int64_t)clip * T * 2 * C + head * 64 + c4 * 4;
    auto put = [&](int r, const float4 &w) {
        const unsigned a = la::x2::pack_hi_lo(w.x * s), b = la::x2::pack_hi_lo(w.y * s), c = la::x2::pack_hi_lo(w.z * s), d = la::x2::pack_hi_lo(w.w * s);
        *reinterpret_cast<uint2 *>(dst + (int64_t)r * 2 * C) = make_uint2((a & 0xffffu) | (b << 16), (c & 0xffffu) | (d << 16));
        *reinterpret_cast<uint2 *>(dst + (int64_t)r * 2 * C + C) = make_uint2((a >> 16) | (b & 0xffff0000u), (c >> 16) | (d & 0xffff0000u));
    };
    if (cached) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int r = r0 + 64 * j;
            if (r < T && j % parts == part) put(r, v[j]);
        }
    } else {
        for (int r = r0, j = 0; r < T; r += 64, ++j)
            if (j % parts == part) put(r, *reinterpret_cast<const float4 *>(src + (int64_t)r * ld));
    }
}

// Block -> (query tile, head, clip): as la_attention.hip block_coord -- the query tiles of one (clip, head) share one XCD's L2
struct BlockCoord { int qt, head, clip; };
__device__ __forceinline__ BlockCoord block_coord(int nq, int n_head, int batch) {
    const int L = blockIdx.x, pairs = n_head * batch;
    int pair, qt;
    if ((pairs & 7) == 0) {
        const int x = L & 7, idx = L >> 3;
        pair = x + 8 * (idx / nq);
        qt = idx % nq;
    } else {
        pair = L / nq;
        qt = L % nq;
    }
    return BlockCoord{qt, pair % n_head, pair / n_head};
}
// accumulator register -> row (key / dv index) inside a 32x32 tile for lane half h
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// K images: 16-B slot s of row r at slot s ^ ((r >> 1) & 7) (ds_read_b128 rows); V images: s ^ (((r >> 1) & 1) << 2) (ds_read_b64_tr_b16)
__device__ __forceinline__ int kswz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int vswz(int r) { return ((r >> 1) & 1) << 2; }
__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// staging: a piece = rows 8i .. 8i+7 of one image (16 B per lane); wave w brings pieces w * PER .. of the hi and lo image of a K or V tile
template <int PER> struct KvOff { unsigned k[PER], v[PER]; };
template <int PER>
__device__ __forceinline__ KvOff<PER> kv_offsets(int64_t C, int key0, int T, int wave, int lane) {
    KvOff<PER> o;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int r = (wave * PER + i) * 8 + (lane >> 3), ps = lane & 7;
        const int rr = key0 + r < T ? r : T - 1 - key0;          // rows past the end: the last key (masked in the scores)
        o.k[i] = (unsigned)(rr * C * 4) + ((ps ^ kswz(r)) << 4);
        o.v[i] = (unsigned)(rr * C * 4) + ((ps ^ vswz(r)) << 4);
    }
    return o;
}
template <int PER>
__device__ __forceinline__ void stage_tile(const unsigned short *base, int64_t C, int key0, const unsigned (&off)[PER], unsigned buf, int wave) {
    const unsigned short *src = base + (int64_t)key0 * 2 * C;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const unsigned piece = (wave * PER + i) * 1024;
        la::glds16_so(off[i], src, buf + piece);
        la::glds16_so(off[i], src + C, buf + IMG + piece);
    }
}

// (a, b) -> packed halves hi = (f16 a, f16 b) and lo = (f16 (a - hi.x), f16 (b - hi.y)): one v_cvt_pk_f16_f32 and two mixed-precision
// fmas that read hi as it lies (v_fma_mixlo / mixhi_f16: (float)hi * -1 + a, rounded to half) instead of convert-back, subtract, convert
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 hv = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(unsigned, hv);
    // one asm statement: the hazard recognizer does not look inside -- a wait state after a transcendental producer of a / b
    // (v_exp_f32), one between the two partial writes of the destination (as hipcc places them around its own v_fma_mix), and two
    // before a matrix instruction may read the result (vector write -> MFMA operand: without them the key sweep's dK came out
    // different from run to run whenever hipcc placed the consuming MFMA right behind)
    asm("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1"
        : "=&v"(lo)
        : "v"(hi), "v"(a), "v"(b));
}

// NW = 8 waves per workgroup, 32 queries each; K / V tiles of 64 keys (hi and lo image each, double-buffered: 64 KiB); 164 VGPRs, two
// waves per SIMD.  Measured at the fine-tune shape (16 clips x 16 heads x 1500; profiles/r5_attention_f16x2_forward.txt): 517 us per
// layer = 34 % of the f16 MFMA peak over the three-fold products, the same fraction of peak the 16-bit inference kernel reaches -- against
// 1310-1360 us of the float32-MFMA kernel.  Forms that did not move it: 128-query workgroups (two per CU); the next tile's score MFMAs
// software-pipelined under the softmax inside the wave (one basic block, MFMA and vector instructions interleaved 1 : 7 by hipcc: 517 us
// again); 32-key halves to fit 128 VGPRs = four waves per SIMD (63 spilled registers: 793 us; the same halves at 168 VGPRs: 586 us).
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attention_x2_fwd_kernel(FwdParams p) {
    constexpr int QT = 32 * NW, PER = 8 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * IMG];      // [buf][K hi | K lo | V hi | V lo]
    const int T = p.kv_len;
    const int64_t C = 64 * (int64_t)p.n_head;
    const BlockCoord bc = block_coord((p.q_len + QT - 1) / QT, p.n_head, p.batch);
    const int qt = bc.qt, head = bc.head, clip = bc.clip;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const unsigned short *kbase = p.kp + (int64_t)clip * T * 2 * C + head * 64;
    const unsigned short *vbase = p.vp + (int64_t)clip * T * 2 * C + head * 64;
    const int bh = clip * p.n_head + head;
    const float kScale = p.sq[bh] * p.sk[bh] * kLog2e;       // raw accumulator -> score in the exp2 domain (powers of two times log2 e)

    // Q fragments (B operand): lane (q = i32, h) holds Q[q][16c + 8h .. +8] of both planes
    int qrow = qt * QT + wave * 32 + i32;
    const bool q_valid = qrow < p.q_len;
    qrow = q_valid ? qrow : p.q_len - 1;
    uint4 qh[4], ql[4];
    {
        const unsigned short *qb = p.qp + ((int64_t)clip * p.q_len + qrow) * 2 * C + head * 64 + 8 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            qh[c] = *reinterpret_cast<const uint4 *>(qb + 16 * c);
            ql[c] = *reinterpret_cast<const uint4 *>(qb + C + 16 * c);
        }
    }
    f32x16 o[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_run = -INFINITY, l_part = 0.f;

    int nkv = (T + KT - 1) / KT;
    const int nkv_all = nkv;
    if (p.causal) nkv = min(nkv, (min(p.q_len, (qt + 1) * QT) - 1) / KT + 1);   // tiles above the block's diagonal are all masked
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(la::lds_addr_u32(lds));
    const KvOff<PER> off_full = kv_offsets<PER>(C, 0, KT, wave, lane);
    const KvOff<PER> off_last = kv_offsets<PER>(C, (nkv_all - 1) * KT, T, wave, lane);
    auto stage = [&](int t, unsigned buf) __attribute__((always_inline)) {
        stage_tile<PER>(kbase, C, t * KT, t == nkv_all - 1 ? off_last.k : off_full.k, buf, wave);
        stage_tile<PER>(vbase, C, t * KT, t == nkv_all - 1 ? off_last.v : off_full.v, buf + 2 * IMG, wave);
    };
    stage(0, lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the Q fragments are "used" here so that their wait sits before the loop (la_attention.hip: otherwise every tile waits for its
    // successor's staging loads)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 t0 = __builtin_bit_cast(u32x4, qh[c]), t1 = __builtin_bit_cast(u32x4, ql[c]);
        asm volatile("" : "+v"(t0), "+v"(t1));
        qh[c] = __builtin_bit_cast(uint4, t0);
        ql[c] = __builtin_bit_cast(uint4, t1);
    }
    __syncthreads();

    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;

    // maskc: the tile holds keys past the end of the clip or, causal, past some query -- an instantiation of its own: as a test inside the
    // tile body the compiler turns the mask into compare / select instructions that every tile executes
    auto tile = [&](int t, auto curc, auto maskc) __attribute__((always_inline)) {
        constexpr int cur = decltype(curc)::value;
        const unsigned char *kl = lds + cur * 4 * IMG;
        const unsigned char *vl = kl + 2 * IMG;
        if (t + 1 < nkv && !(LA_X2F_KO & 8)) stage(t + 1, lds0 + (cur ^ 1) * 4 * IMG);
        // ---- S^T = K Q^T: two 32-key sub-tiles, three products each (small terms first); all sixteen K fragments requested before the
        // first MFMA (64 VGPRs): the reads return under the MFMAs ----
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
        {
            uint4 kfh[2][4], kfl[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int row = sub * 32 + i32;
                    const int off = row * 128 + (((2 * c + h) ^ kswz(row)) << 4);
                    if constexpr (LA_X2F_KO & 4) { kfh[sub][c] = qh[(c + sub) & 3]; kfl[sub][c] = ql[(c + sub) & 3]; }
                    else {
                        kfh[sub][c] = *reinterpret_cast<const uint4 *>(kl + off);
                        kfl[sub][c] = *reinterpret_cast<const uint4 *>(kl + IMG + off);
                    }
                }
            if constexpr (LA_X2F_KO & 16) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) s[sub][c] += __builtin_bit_cast(float, kfh[sub][c].x) + __builtin_bit_cast(float, kfl[sub][c].w);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) s[sub] = mfma(__builtin_bit_cast(f16x8, kfl[sub][c]), __builtin_bit_cast(f16x8, qh[c]), s[sub]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) s[sub] = mfma(__builtin_bit_cast(f16x8, kfh[sub][c]), __builtin_bit_cast(f16x8, ql[c]), s[sub]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) s[sub] = mfma(__builtin_bit_cast(f16x8, kfh[sub][c]), __builtin_bit_cast(f16x8, qh[c]), s[sub]);
            }
        }
        if constexpr (decltype(maskc)::value) {
            const int kmax = p.causal ? min(T - 1, qrow) : T - 1;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KT + sub * 32 + acc_row(r, h) > kmax) s[sub][r] = -INFINITY;
        }
        // ---- online softmax in the exp2 domain; P carries 2^13 ----
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * kScale;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // exp2(-inf) = 0 on the first tile
        m_run = m_new;
        const float mneg = 13.0f - m_new;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if constexpr (LA_X2F_KO & 1) {
                    s[sub][r] = fmaf(s[sub][r], kScale, mneg);
                    s[sub][r + 1] = fmaf(s[sub][r + 1], kScale, mneg);
                } else {
                    s[sub][r] = __builtin_amdgcn_exp2f(fmaf(s[sub][r], kScale, mneg));
                    s[sub][r + 1] = __builtin_amdgcn_exp2f(fmaf(s[sub][r + 1], kScale, mneg));
                }
                ps0 += s[sub][r];
                ps1 += s[sub][r + 1];
            }
        l_part = l_part * alpha + (ps0 + ps1);
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {             // wave-uniform; exact: alpha == 1 changes nothing
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // P of 16 keys as B operand (hi, lo): element j <-> accumulator register 8 ks + j; split here, under the previous MFMAs
                uint4 ph, pl;
                split_pair(s[sub][8 * ks + 0], s[sub][8 * ks + 1], ph.x, pl.x);
                split_pair(s[sub][8 * ks + 2], s[sub][8 * ks + 3], ph.y, pl.y);
                split_pair(s[sub][8 * ks + 4], s[sub][8 * ks + 5], ph.z, pl.z);
                split_pair(s[sub][8 * ks + 6], s[sub][8 * ks + 7], ph.w, pl.w);
                const f16x8 phv = __builtin_bit_cast(f16x8, ph), plv = __builtin_bit_cast(f16x8, pl);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // lane (dv = 32b + 16(g&1) + (lane&15), half g>>1): 4 keys key0 .. key0+3 per read
                    const int key0 = sub * 32 + 16 * ks + 4 * (g >> 1);
                    const int slot = b * 4 + 2 * (g & 1) + (pp >> 1);
                    const int r0 = key0 + q4, r1 = key0 + 8 + q4;
                    const int a0 = r0 * 128 + ((slot ^ vswz(r0)) << 4) + (pp & 1) * 8, a1 = r1 * 128 + ((slot ^ vswz(r1)) << 4) + (pp & 1) * 8;
                    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
                    f16x8 vh, vlo;
                    if constexpr (LA_X2F_KO & 2) {
                        vh = __builtin_bit_cast(f16x8, qh[(2 * sub + ks + b) & 3]);
                        vlo = __builtin_bit_cast(f16x8, ql[(2 * sub + ks + b) & 3]);
                    } else {
                        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + a0)), h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + a1));
                        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + IMG + a0)), l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + IMG + a1));
                        vh = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                        vlo = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    if constexpr (LA_X2F_KO & 16) {
                        o[b][2 * sub + ks] += (float)vh[0] + (float)vlo[7] + (float)phv[0] + (float)plv[7];
                    } else {
                        o[b] = mfma(vlo, phv, o[b]);
                        o[b] = mfma(vh, plv, o[b]);
                        o[b] = mfma(vh, phv, o[b]);
                    }
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(LA_X2F_KO & 32)) __syncthreads();
    };
    auto step = [&](int t, auto curc) __attribute__((always_inline)) {
        if (__builtin_expect((t + 1) * KT > T || p.causal, 0)) tile(t, curc, std::true_type{});
        else tile(t, curc, std::false_type{});
    };
    for (int t = 0; t < nkv; t += 2) {
        step(t, std::integral_constant<int, 0>{});
        if (t + 1 < nkv) step(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: O[q][dv] = O^T sv / l; lse = m ln 2 + ln(l / 2^13) ----
    const float l = l_part + __shfl_xor(l_part, 32);
    const float inv = p.sv[bh] / l;
    if (p.lse && q_valid && h == 0) p.lse[((int64_t)clip * p.n_head + head) * p.q_len + qrow] = fmaf(m_run - 13.0f, kLn2, __logf(l));
    if (q_valid) {
        float *orow = p.out + ((int64_t)clip * p.q_len + qrow) * p.ld_out + head * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4 *>(orow + 32 * b + 8 * r4 + 4 * h) =
                    make_float4(o[b][4 * r4 + 0] * inv, o[b][4 * r4 + 1] * inv, o[b][4 * r4 + 2] * inv, o[b][4 * r4 + 3] * inv);
    }
}

// =======================================================================================================================================
// Backward: dQ, dK, dV of O = softmax(Q K^T) V from dO (whisper/model.py MultiHeadAttention.qkv_attention under train_multitask.py:325-326
// `loss.backward()`), the two sweeps of la_attention_bwd.hip (no atomics, nothing of size Tq x Tk leaves the CU) in the register-resident
// form of the forward above, every product three f16 MFMAs:
//   key sweep    a wave owns 32 keys -- K, V fragments (B operands) and the dV^T, dK^T accumulators live in registers -- and walks the query
//                tiles (launched as a dV sweep and a dK sweep, see attention_x2_bwd_kv_kernel):  S = Q K^T,  dP = dO V^T  (A operands: row fragments of the Q / dO images),  P = exp(S - lse),
//                dS = P o (dP - D),  dV^T += dO^T P,  dK^T += Q^T dS  (A operands: transposed fragments, ds_read_b64_tr_b16)
//   query sweep  a wave owns 32 queries -- Q, dO fragments and dQ^T -- and walks the key tiles:  S^T = K Q^T,  dP^T = V dO^T,
//                dS^T = P^T o (dP^T - D),  dQ^T += K^T dS^T
// A lane holds one key's (query's) column of the score tile, so P and dS leave the accumulators as the next product's B operand.
// Operands: q, k, v, dO as half planes with one power-of-two scale per (clip, head) (heads_split_kernel).  P is split with the scale 2^13
// (folded into the exponent).  dS is split with a scale from a bound that needs no reduction: |dS_ij| <= |dP_ij| + |D_i| <=
// 2 |dO_i| |v_j| <= 128 max|dO| max|v| < 2^35 / (s_dO s_v), so dS 2^-20 s_dO s_v < 2^15 -- in terms of the raw dP accumulator
// (= dP s_dO s_v) that is  P (dP_acc 2^-20 - D 2^-20 s_dO s_v); entries far below the bound lose relative, never absolute, precision
// (half's subnormal range: 2^-39 of the bound).  An image that is read both ways (rows for one product, transposed for another) is staged
// twice, once per swizzle: LDS 128 KiB (key sweep) / 96 KiB (query sweep), one workgroup of eight waves per CU.
struct BwdParams {
    const unsigned short *qp, *kp, *vp, *dop;    // planes [token][2][C]
    const float *sq, *sk, *sv, *sdo;             // inverse scales [batch][n_head]
    const float *lse, *dvec;                     // [batch][n_head][q_len]
    float *dq, *dk, *dv;
    int64_t ld_dq, ld_dkv;
    int q_len, kv_len, n_head, causal, batch;
};

// fragment readers.  Row fragment (A operand, M = token row of the image, K = 16 columns of the head dimension: c): ds_read_b128
__device__ __forceinline__ f16x8 row_frag(const unsigned char *img, int row, int c, int h) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(img + row * 128 + (((2 * c + h) ^ kswz(row)) << 4)));
}
// Transposed fragment (A operand, M = 32 columns of the head dimension: b, K = 16 token rows tok0 ..): lane (col = 32b + 16(g&1) + (lane&15),
// half g>>1) gets token rows tok0 + 4(g>>1) + {0..3} and + 8
struct TrLane { int g, q4, pp; };
__device__ __forceinline__ f16x8 tr_frag(const unsigned char *img, int tok0, int b, const TrLane &L) {
    const int key0 = tok0 + 4 * (L.g >> 1);
    const int slot = b * 4 + 2 * (L.g & 1) + (L.pp >> 1);
    const int r0 = key0 + L.q4, r1 = key0 + 8 + L.q4;
    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + r0 * 128 + ((slot ^ vswz(r0)) << 4) + (L.pp & 1) * 8));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + r1 * 128 + ((slot ^ vswz(r1)) << 4) + (L.pp & 1) * 8));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}
// acc += A B with both operands as (hi, lo): small terms first
__device__ __forceinline__ f32x16 mma3(f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl, f32x16 acc) {
    acc = mfma(al, bh, acc);
    acc = mfma(ah, bl, acc);
    return mfma(ah, bh, acc);
}
// eight consecutive accumulator registers (a K-step of 16 token rows) -> B operand (hi, lo)
__device__ __forceinline__ void split8(const f32x16 &s, int ks, f16x8 &hi, f16x8 &lo) {
    uint4 ph, pl;
    split_pair(s[8 * ks + 0], s[8 * ks + 1], ph.x, pl.x);
    split_pair(s[8 * ks + 2], s[8 * ks + 3], ph.y, pl.y);
    split_pair(s[8 * ks + 4], s[8 * ks + 5], ph.z, pl.z);
    split_pair(s[8 * ks + 6], s[8 * ks + 7], ph.w, pl.w);
    hi = __builtin_bit_cast(f16x8, ph);
    lo = __builtin_bit_cast(f16x8, pl);
}

constexpr int BWD_NW = 8;
constexpr int KV_BUF = 8 * IMG;                 // Q rows hi|lo, Q transposed hi|lo, dO rows hi|lo, dO transposed hi|lo
constexpr int KV_LDS = 2 * KV_BUF + 2 * 2 * 64 * 4;
constexpr int Q_BUF = 6 * IMG;                  // K rows hi|lo, K transposed hi|lo, V rows hi|lo
constexpr int Q_LDS = 2 * Q_BUF;

// MODE 1: dV alone (K fragments, dV^T, scores); MODE 2: dK alone (K, V fragments, dK^T, scores, dP); MODE 3: both in one sweep.  The
// library launches the PAIR: one sweep for both holds 160 resident registers + fragments and hipcc spills 63 of them (1359 us per layer of
// 16 clips); the pair recomputes the scores once more (60 instead of 48 MFMAs per 32 queries), stages 4 + 6 instead of 8 images per tile
// and does not spill: 499 + 720 us (profiles/r5_attention_f16x2_backward.txt).
template <int MODE>
__global__ __launch_bounds__(64 * BWD_NW, 2) void attention_x2_bwd_kv_kernel(BwdParams p) {
    constexpr bool DV = MODE & 1, DK = MODE & 2;
    constexpr int NW = BWD_NW, KB = 32 * NW, PER = 8 / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *stats = reinterpret_cast<float *>(lds + 2 * KV_BUF);      // [buf][nl | ds][64]
    const int Tq = p.q_len, Tk = p.kv_len;
    const int64_t C = 64 * (int64_t)p.n_head;
    const BlockCoord bc = block_coord((Tk + KB - 1) / KB, p.n_head, p.batch);
    const int kt = bc.qt, head = bc.head, clip = bc.clip;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const TrLane L{lane >> 4, (lane & 15) >> 2, lane & 3};
    const unsigned short *qbase = p.qp + (int64_t)clip * Tq * 2 * C + head * 64;
    const unsigned short *dobase = p.dop + (int64_t)clip * Tq * 2 * C + head * 64;
    const int bh = clip * p.n_head + head;
    const float isq = p.sq[bh], isk = p.sk[bh], isv = p.sv[bh], isdo = p.sdo[bh];
    const float kScale = isq * isk * kLog2e;
    const float dscale = 0x1p-33f / (isdo * isv);          // D -> D s_dS 2^-13 (P carries 2^13)
    const float *lse_b = p.lse + (int64_t)bh * Tq, *dvec_b = p.dvec + (int64_t)bh * Tq;

    // K and V fragments (B operands): lane (key = i32, h) holds K[key][16c + 8h .. +8] of both planes
    const int krow_true = kt * KB + wave * 32 + i32;
    const bool k_valid = krow_true < Tk;
    const int krow = k_valid ? krow_true : Tk - 1;
    uint4 kh[4], kl[4], vh[4], vl[4];
    {
        const unsigned short *kb = p.kp + ((int64_t)clip * Tk + krow) * 2 * C + head * 64 + 8 * h;
        const unsigned short *vb = p.vp + ((int64_t)clip * Tk + krow) * 2 * C + head * 64 + 8 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kh[c] = *reinterpret_cast<const uint4 *>(kb + 16 * c);
            kl[c] = *reinterpret_cast<const uint4 *>(kb + C + 16 * c);
            if constexpr (DK) {
                vh[c] = *reinterpret_cast<const uint4 *>(vb + 16 * c);
                vl[c] = *reinterpret_cast<const uint4 *>(vb + C + 16 * c);
            } else {
                vh[c] = vl[c] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    f32x16 dv[2], dk[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dv[b][r] = 0.f; dk[b][r] = 0.f; }

    const int nq = (Tq + KT - 1) / KT;
    const int t0 = p.causal ? (kt * KB) / KT : 0;            // causal: query tiles wholly before the block's first key see none of its keys
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(la::lds_addr_u32(lds));
    const KvOff<PER> off_full = kv_offsets<PER>(C, 0, KT, wave, lane);
    const KvOff<PER> off_last = kv_offsets<PER>(C, (nq - 1) * KT, Tq, wave, lane);
    auto stage = [&](int t, unsigned buf) __attribute__((always_inline)) {
        const KvOff<PER> &o = t == nq - 1 ? off_last : off_full;
        stage_tile<PER>(qbase, C, t * KT, o.k, buf, wave);
        if constexpr (DK) stage_tile<PER>(qbase, C, t * KT, o.v, buf + 2 * IMG, wave);
        if constexpr (DK) stage_tile<PER>(dobase, C, t * KT, o.k, buf + 4 * IMG, wave);
        if constexpr (DV) stage_tile<PER>(dobase, C, t * KT, o.v, buf + 6 * IMG, wave);
    };
    // row statistics of a query tile: threads 0..63 bring 13 - lse log2 e, threads 64..127 D s_dS 2^-13 (rows past the end: 0, masked below)
    auto stat_load = [&](int t) -> float {
        float v = 0.f;
        if (tid < 128) {
            const int q = t * KT + (tid & 63);
            if (q < Tq) v = tid < 64 ? 13.0f - lse_b[q] * kLog2e : (DK ? dvec_b[q] * dscale : 0.f);
        }
        return v;
    };
    if (t0 < nq) {
        stage(t0, lds0);
        const float sv0 = stat_load(t0);
        if (tid < 128) stats[tid] = sv0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {          // (fragment loads waited for here, not inside the loop: see the forward kernel)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 a0 = __builtin_bit_cast(u32x4, kh[c]), a1 = __builtin_bit_cast(u32x4, kl[c]), a2 = __builtin_bit_cast(u32x4, vh[c]), a3 = __builtin_bit_cast(u32x4, vl[c]);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        kh[c] = __builtin_bit_cast(uint4, a0); kl[c] = __builtin_bit_cast(uint4, a1); vh[c] = __builtin_bit_cast(uint4, a2); vl[c] = __builtin_bit_cast(uint4, a3);
    }
    __syncthreads();

    auto tile = [&](int t, auto curc, auto maskc) __attribute__((always_inline)) {
        constexpr int cur = decltype(curc)::value;
        const unsigned char *qr = lds + cur * KV_BUF, *qtr = qr + 2 * IMG, *dor = qr + 4 * IMG, *dotr = qr + 6 * IMG;
        const float *st = stats + cur * 128;
        float sv_next = 0.f;
        if (t + 1 < nq) {
            stage(t + 1, lds0 + (cur ^ 1) * KV_BUF);
            sv_next = stat_load(t + 1);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int row = sub * 32 + i32;
            // ---- S = Q K^T and dP = dO V^T for 32 queries x this wave's 32 keys (lane = key, registers = queries) ----
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 ah = row_frag(qr, row, c, h), al = row_frag(qr + IMG, row, c, h);
                s = mma3(ah, al, __builtin_bit_cast(f16x8, kh[c]), __builtin_bit_cast(f16x8, kl[c]), s);
                if constexpr (DK) {
                    const f16x8 gh = row_frag(dor, row, c, h), gl = row_frag(dor + IMG, row, c, h);
                    dp = mma3(gh, gl, __builtin_bit_cast(f16x8, vh[c]), __builtin_bit_cast(f16x8, vl[c]), dp);
                }
            }
            // ---- P 2^13 = exp2(s - lse + 13) (in s), dS s_dS = P 2^13 (dP_acc 2^-33 - D s_dS 2^-13) (in dp); rows = queries ----
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 nl = *reinterpret_cast<const float4 *>(st + sub * 32 + 8 * r4 + 4 * h);
                const float4 ds = DK ? *reinterpret_cast<const float4 *>(st + 64 + sub * 32 + 8 * r4 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float nlv[4] = {nl.x, nl.y, nl.z, nl.w}, dsv[4] = {ds.x, ds.y, ds.z, ds.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * r4 + e;
                    float a = fmaf(s[r], kScale, nlv[e]);
                    if constexpr (decltype(maskc)::value) {
                        const int q = t * KT + sub * 32 + 8 * r4 + 4 * h + e;
                        if (q >= Tq || (p.causal && krow_true > q)) a = -INFINITY;
                    }
                    const float pv = __builtin_amdgcn_exp2f(a);
                    s[r] = pv;
                    if constexpr (DK) dp[r] = pv * fmaf(dp[r], 0x1p-33f, -dsv[e]);
                }
            }
            // ---- dV^T += dO^T P,  dK^T += Q^T dS ----
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ph, pl, sh, sl;
                if constexpr (DV) split8(s, ks, ph, pl);
                if constexpr (DK) split8(dp, ks, sh, sl);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if constexpr (DV) dv[b] = mma3(tr_frag(dotr, sub * 32 + 16 * ks, b, L), tr_frag(dotr + IMG, sub * 32 + 16 * ks, b, L), ph, pl, dv[b]);
                    if constexpr (DK) dk[b] = mma3(tr_frag(qtr, sub * 32 + 16 * ks, b, L), tr_frag(qtr + IMG, sub * 32 + 16 * ks, b, L), sh, sl, dk[b]);
                }
            }
        }
        if (t + 1 < nq && tid < 128) stats[(cur ^ 1) * 128 + tid] = sv_next;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto step = [&](int t, auto curc) __attribute__((always_inline)) {
        // masked: the tile holds rows past the last query or, causal, queries before some key of the block
        if (__builtin_expect((t + 1) * KT > Tq || (p.causal && t * KT < kt * KB + KB), 0)) tile(t, curc, std::true_type{});
        else tile(t, curc, std::false_type{});
    };
    for (int t = t0; t < nq; t += 2) {
        step(t, std::integral_constant<int, 0>{});
        if (t + 1 < nq) step(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: dV = dV^T s_dO^-1 2^-13,  dK = dK^T s_q^-1 s_dS^-1 (s_dS = 2^-20 s_dO s_v in terms of the scales) ----
    if (k_valid) {
        const float cv = isdo * 0x1p-13f, ck = isq * 0x1p20f * isdo * isv;
        float *dvrow = p.dv + ((int64_t)clip * Tk + krow) * p.ld_dkv + head * 64;
        float *dkrow = p.dk + ((int64_t)clip * Tk + krow) * p.ld_dkv + head * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if constexpr (DV)
                    *reinterpret_cast<float4 *>(dvrow + 32 * b + 8 * r4 + 4 * h) =
                        make_float4(dv[b][4 * r4 + 0] * cv, dv[b][4 * r4 + 1] * cv, dv[b][4 * r4 + 2] * cv, dv[b][4 * r4 + 3] * cv);
                if constexpr (DK)
                    *reinterpret_cast<float4 *>(dkrow + 32 * b + 8 * r4 + 4 * h) =
                        make_float4(dk[b][4 * r4 + 0] * ck, dk[b][4 * r4 + 1] * ck, dk[b][4 * r4 + 2] * ck, dk[b][4 * r4 + 3] * ck);
            }
    }
}

__global__ __launch_bounds__(64 * BWD_NW, 2) void attention_x2_bwd_q_kernel(BwdParams p) {
    constexpr int NW = BWD_NW, QT = 32 * NW, PER = 8 / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int Tq = p.q_len, T = p.kv_len;
    const int64_t C = 64 * (int64_t)p.n_head;
    const BlockCoord bc = block_coord((Tq + QT - 1) / QT, p.n_head, p.batch);
    const int qt = bc.qt, head = bc.head, clip = bc.clip;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const TrLane L{lane >> 4, (lane & 15) >> 2, lane & 3};
    const unsigned short *kbase = p.kp + (int64_t)clip * T * 2 * C + head * 64;
    const unsigned short *vbase = p.vp + (int64_t)clip * T * 2 * C + head * 64;
    const int bh = clip * p.n_head + head;
    const float isq = p.sq[bh], isk = p.sk[bh], isv = p.sv[bh], isdo = p.sdo[bh];
    const float kScale = isq * isk * kLog2e;

    int qrow = qt * QT + wave * 32 + i32;
    const bool q_valid = qrow < Tq;
    qrow = q_valid ? qrow : Tq - 1;
    uint4 qh[4], ql[4], gh[4], gl[4];
    {
        const unsigned short *qb = p.qp + ((int64_t)clip * Tq + qrow) * 2 * C + head * 64 + 8 * h;
        const unsigned short *gb = p.dop + ((int64_t)clip * Tq + qrow) * 2 * C + head * 64 + 8 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            qh[c] = *reinterpret_cast<const uint4 *>(qb + 16 * c);
            ql[c] = *reinterpret_cast<const uint4 *>(qb + C + 16 * c);
            gh[c] = *reinterpret_cast<const uint4 *>(gb + 16 * c);
            gl[c] = *reinterpret_cast<const uint4 *>(gb + C + 16 * c);
        }
    }
    const float nl = 13.0f - p.lse[(int64_t)bh * Tq + qrow] * kLog2e;
    const float dsq = p.dvec[(int64_t)bh * Tq + qrow] * (0x1p-33f / (isdo * isv));
    f32x16 dq[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[b][r] = 0.f;

    int nkv = (T + KT - 1) / KT;
    const int nkv_all = nkv;
    if (p.causal) nkv = min(nkv, (min(Tq, (qt + 1) * QT) - 1) / KT + 1);   // tiles above the block's diagonal are all masked
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(la::lds_addr_u32(lds));
    const KvOff<PER> off_full = kv_offsets<PER>(C, 0, KT, wave, lane);
    const KvOff<PER> off_last = kv_offsets<PER>(C, (nkv_all - 1) * KT, T, wave, lane);
    auto stage = [&](int t, unsigned buf) __attribute__((always_inline)) {
        const KvOff<PER> &o = t == nkv_all - 1 ? off_last : off_full;
        stage_tile<PER>(kbase, C, t * KT, o.k, buf, wave);
        stage_tile<PER>(kbase, C, t * KT, o.v, buf + 2 * IMG, wave);
        stage_tile<PER>(vbase, C, t * KT, o.k, buf + 4 * IMG, wave);
    };
    stage(0, lds0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 a0 = __builtin_bit_cast(u32x4, qh[c]), a1 = __builtin_bit_cast(u32x4, ql[c]), a2 = __builtin_bit_cast(u32x4, gh[c]), a3 = __builtin_bit_cast(u32x4, gl[c]);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        qh[c] = __builtin_bit_cast(uint4, a0); ql[c] = __builtin_bit_cast(uint4, a1); gh[c] = __builtin_bit_cast(uint4, a2); gl[c] = __builtin_bit_cast(uint4, a3);
    }
    __syncthreads();

    auto tile = [&](int t, auto curc, auto maskc) __attribute__((always_inline)) {
        constexpr int cur = decltype(curc)::value;
        const unsigned char *kr = lds + cur * Q_BUF, *ktr = kr + 2 * IMG, *vr = kr + 4 * IMG;
        if (t + 1 < nkv) stage(t + 1, lds0 + (cur ^ 1) * Q_BUF);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int row = sub * 32 + i32;
            // ---- S^T = K Q^T and dP^T = V dO^T for 32 keys x this wave's 32 queries (lane = query, registers = keys) ----
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 ah = row_frag(kr, row, c, h), al = row_frag(kr + IMG, row, c, h);
                s = mma3(ah, al, __builtin_bit_cast(f16x8, qh[c]), __builtin_bit_cast(f16x8, ql[c]), s);
                const f16x8 bh_ = row_frag(vr, row, c, h), bl_ = row_frag(vr + IMG, row, c, h);
                dp = mma3(bh_, bl_, __builtin_bit_cast(f16x8, gh[c]), __builtin_bit_cast(f16x8, gl[c]), dp);
            }
            const int kmax = p.causal ? min(T - 1, qrow) : T - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = fmaf(s[r], kScale, nl);
                if constexpr (decltype(maskc)::value)
                    if (t * KT + sub * 32 + acc_row(r, h) > kmax) a = -INFINITY;
                const float pv = __builtin_amdgcn_exp2f(a);
                dp[r] = pv * fmaf(dp[r], 0x1p-33f, -dsq);
            }
            // ---- dQ^T += K^T dS^T ----
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 sh, sl;
                split8(dp, ks, sh, sl);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    dq[b] = mma3(tr_frag(ktr, sub * 32 + 16 * ks, b, L), tr_frag(ktr + IMG, sub * 32 + 16 * ks, b, L), sh, sl, dq[b]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto step = [&](int t, auto curc) __attribute__((always_inline)) {
        if (__builtin_expect((t + 1) * KT > T || p.causal, 0)) tile(t, curc, std::true_type{});
        else tile(t, curc, std::false_type{});
    };
    for (int t = 0; t < nkv; t += 2) {
        step(t, std::integral_constant<int, 0>{});
        if (t + 1 < nkv) step(t + 1, std::integral_constant<int, 1>{});
    }

    if (q_valid) {
        const float cq = isk * 0x1p20f * isdo * isv;
        float *dqrow = p.dq + ((int64_t)clip * Tq + qrow) * p.ld_dq + head * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4 *>(dqrow + 32 * b + 8 * r4 + 4 * h) =
                    make_float4(dq[b][4 * r4 + 0] * cq, dq[b][4 * r4 + 1] * cq, dq[b][4 * r4 + 2] * cq, dq[b][4 * r4 + 3] * cq);
    }
}

// workgroups per (clip, head) slice of the operand split: enough to fill the chip when the batch is small
static dim3 heads_split_grid(int n_head, int batch) {
    int parts = 1;
    while (parts < 8 && n_head * batch * parts < 256) parts *= 2;
    return dim3(n_head, batch, parts);
}

// D_i = sum_d dO_id O_id per (clip, head, query): one 16-lane group per row and head (the statistics launch of la_attention_bwd_f32 also
// recomputes lse when the forward did not hand it over; with lse at hand this is all that is left of it)
__global__ __launch_bounds__(256) void attn_dvec_kernel(const float *o, int64_t ld_o, const float *dout, int64_t ld_do, int B, int Tq, int H, float *dvec) {
    const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);          // (row, head)
    const int part = threadIdx.x & 15;
    const int64_t rows = (int64_t)B * Tq;
    if (item >= rows * H) return;
    const int64_t row = item / H;
    const int head = (int)(item % H);
    const float4 a = *reinterpret_cast<const float4 *>(o + row * ld_o + head * 64 + part * 4);
    const float4 g = *reinterpret_cast<const float4 *>(dout + row * ld_do + head * 64 + part * 4);
    float s = (a.x * g.x + a.y * g.y) + (a.z * g.z + a.w * g.w);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
    if (part == 0) dvec[((int64_t)(row / Tq) * H + head) * Tq + row % Tq] = s;
}

struct FwdWorkspace {
    unsigned short *qp, *kp, *vp;
    float *sq, *sk, *sv;
    size_t bytes;
};
FwdWorkspace fwd_workspace(void *base, int B, int Tq, int Tk, int H) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t C = 64 * (size_t)H;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += up(n); return o; };
    const size_t oq = take((size_t)B * Tq * 2 * C * 2), ok = take((size_t)B * Tk * 2 * C * 2), ov = take((size_t)B * Tk * 2 * C * 2);
    const size_t osq = take((size_t)B * H * 4), osk = take((size_t)B * H * 4), osv = take((size_t)B * H * 4);
    char *b = reinterpret_cast<char *>(base);
    return FwdWorkspace{reinterpret_cast<unsigned short *>(b + oq), reinterpret_cast<unsigned short *>(b + ok), reinterpret_cast<unsigned short *>(b + ov),
                        reinterpret_cast<float *>(b + osq), reinterpret_cast<float *>(b + osk), reinterpret_cast<float *>(b + osv), off};
}

}  // namespace

extern "C" int la_attention_f16x2_workspace_bytes(int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch >= 0 && q_len >= 0 && kv_len >= 0 && n_head > 0, "attention_f16x2_workspace_bytes: bad arguments");
    *bytes = fwd_workspace(nullptr, batch, q_len, kv_len, n_head).bytes;
    return LA_OK;
}

// la_attention_lse_f32 (la_attention.hip) with the two products on the f16 matrix pipe: same arguments and results (out, lse) to float32
// accuracy, plus a workspace of la_attention_f16x2_workspace_bytes bytes (256-byte aligned) for the operand planes.
extern "C" int la_attention_lse_f16x2(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, float *out, int64_t ld_out,
                                      int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, float *lse, void *workspace,
                                      size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || q_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && out && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_lse_f16x2: bad arguments");
    LA_CHECK_ARG(!causal || q_len == kv_len, "attention_lse_f16x2: causal masking is defined for self-attention (q_len == kv_len)");
    LA_CHECK_ARG(ld_q >= n_head * 64 && ld_kv >= n_head * 64 && ld_out >= n_head * 64, "attention_lse_f16x2: leading dimensions too small");
    LA_CHECK_ARG(ld_q % 4 == 0 && ld_kv % 4 == 0 && ld_out % 4 == 0 && (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 &&
                     (uintptr_t)out % 16 == 0, "attention_lse_f16x2: rows must be 16-byte aligned");
    const FwdWorkspace ws = fwd_workspace(workspace, batch, q_len, kv_len, n_head);
    LA_CHECK_ARG(workspace && (uintptr_t)workspace % 256 == 0 && workspace_bytes >= ws.bytes, "attention_lse_f16x2: workspace missing, misaligned or too small");
    la::TimerScope ts("attention_f16x2", stream);
    const dim3 sgrid = heads_split_grid(n_head, batch);
    hipLaunchKernelGGL(heads_split_kernel, sgrid, dim3(1024), 0, stream, q, ld_q, q_len, n_head, ws.qp, ws.sq);
    hipLaunchKernelGGL(heads_split_kernel, sgrid, dim3(1024), 0, stream, k, ld_kv, kv_len, n_head, ws.kp, ws.sk);
    hipLaunchKernelGGL(heads_split_kernel, sgrid, dim3(1024), 0, stream, v, ld_kv, kv_len, n_head, ws.vp, ws.sv);
    FwdParams p{ws.qp, ws.kp, ws.vp, ws.sq, ws.sk, ws.sv, out, ld_out, lse, q_len, kv_len, n_head, causal ? 1 : 0, batch};
    hipLaunchKernelGGL((attention_x2_fwd_kernel<8>), dim3(la::cdiv(q_len, 256) * n_head * batch), dim3(512), 0, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

namespace {
struct BwdWorkspace {
    unsigned short *qp, *kp, *vp, *dop;
    float *sq, *sk, *sv, *sdo, *lse, *dvec;
    size_t bytes;
};
BwdWorkspace bwd_workspace(void *base, int B, int Tq, int Tk, int H) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t C = 64 * (size_t)H;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += up(n); return o; };
    const size_t oq = take((size_t)B * Tq * 2 * C * 2), ok = take((size_t)B * Tk * 2 * C * 2), ov = take((size_t)B * Tk * 2 * C * 2),
                 og = take((size_t)B * Tq * 2 * C * 2);
    const size_t osq = take((size_t)B * H * 4), osk = take((size_t)B * H * 4), osv = take((size_t)B * H * 4), osg = take((size_t)B * H * 4);
    const size_t ol = take((size_t)B * H * Tq * 4), od = take((size_t)B * H * Tq * 4);
    char *b = reinterpret_cast<char *>(base);
    auto us = [&](size_t o) { return reinterpret_cast<unsigned short *>(b + o); };
    auto fl = [&](size_t o) { return reinterpret_cast<float *>(b + o); };
    return BwdWorkspace{us(oq), us(ok), us(ov), us(og), fl(osq), fl(osk), fl(osv), fl(osg), fl(ol), fl(od), off};
}
}  // namespace

extern "C" int la_attention_bwd_stats_f32(const float *q, int64_t ld_q, const float *k, int64_t ld_kv, const float *o, int64_t ld_o, const float *dout,
                                          int64_t ld_do, int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal,
                                          const float *lse_in, float *lse, float *dvec, void *stream);

extern "C" int la_attention_bwd_f16x2_workspace_bytes(int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch >= 0 && q_len >= 0 && kv_len >= 0 && n_head > 0, "attention_bwd_f16x2_workspace_bytes: bad arguments");
    *bytes = bwd_workspace(nullptr, batch, q_len, kv_len, n_head).bytes;
    return LA_OK;
}

// la_attention_bwd_f32 (la_attention_bwd.hip) with its seven products on the f16 matrix pipe: same arguments and results (dq, dk, dv) to
// float32 accuracy; workspace (256-byte aligned) of la_attention_bwd_f16x2_workspace_bytes bytes for the operand planes and row statistics.
extern "C" int la_attention_bwd_f16x2(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, const float *o, int64_t ld_o,
                                      const float *dout, int64_t ld_do, float *dq, int64_t ld_dq, float *dk, float *dv, int64_t ld_dkv,
                                      int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, const float *lse,
                                      void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || q_len == 0 || kv_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && o && dout && dq && dk && dv && workspace, "attention_bwd_f16x2: null pointer");
    LA_CHECK_ARG(batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_bwd_f16x2: bad sizes");
    LA_CHECK_ARG(!causal || q_len == kv_len, "attention_bwd_f16x2: the causal mask needs q_len == kv_len");
    LA_CHECK_ARG(ld_q % 4 == 0 && ld_kv % 4 == 0 && ld_o % 4 == 0 && ld_do % 4 == 0 && ld_dq % 4 == 0 && ld_dkv % 4 == 0 &&
                     ld_q >= n_head * 64 && ld_kv >= n_head * 64 && ld_o >= n_head * 64 && ld_do >= n_head * 64 && ld_dq >= n_head * 64 &&
                     ld_dkv >= n_head * 64, "attention_bwd_f16x2: leading dimensions must be multiples of 4 and at least n_head * 64");
    LA_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0,
                 "attention_bwd_f16x2: operands must be 16-byte aligned");
    const BwdWorkspace ws = bwd_workspace(workspace, batch, q_len, kv_len, n_head);
    LA_CHECK_ARG((uintptr_t)workspace % 256 == 0 && workspace_bytes >= ws.bytes, "attention_bwd_f16x2: workspace misaligned or too small");
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_x2_bwd_kv_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, KV_LDS));
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_x2_bwd_kv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, KV_LDS));
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_x2_bwd_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS));
        attr_once.mark();
    }
    // row statistics: D = sum dO o O per query (and lse unless the forward handed it over)
    if (lse) {
        const int64_t items = (int64_t)batch * q_len * n_head;
        hipLaunchKernelGGL(attn_dvec_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, stream, o, ld_o, dout, ld_do, batch, q_len, n_head, ws.dvec);
    } else {
        const int rc = la_attention_bwd_stats_f32(q, ld_q, k, ld_kv, o, ld_o, dout, ld_do, batch, q_len, kv_len, n_head, causal, lse, ws.lse, ws.dvec, stream_);
        if (rc != LA_OK) return rc;
    }
    la::TimerScope ts("attention_bwd_f16x2", stream);
    const dim3 sg = heads_split_grid(n_head, batch), sb(1024);
    hipLaunchKernelGGL(heads_split_kernel, sg, sb, 0, stream, q, ld_q, q_len, n_head, ws.qp, ws.sq);
    hipLaunchKernelGGL(heads_split_kernel, sg, sb, 0, stream, k, ld_kv, kv_len, n_head, ws.kp, ws.sk);
    hipLaunchKernelGGL(heads_split_kernel, sg, sb, 0, stream, v, ld_kv, kv_len, n_head, ws.vp, ws.sv);
    hipLaunchKernelGGL(heads_split_kernel, sg, sb, 0, stream, dout, ld_do, q_len, n_head, ws.dop, ws.sdo);
    BwdParams p{ws.qp, ws.kp, ws.vp, ws.dop, ws.sq, ws.sk, ws.sv, ws.sdo, lse ? lse : ws.lse, ws.dvec, dq, dk, dv, ld_dq, ld_dkv, q_len, kv_len, n_head, causal ? 1 : 0, batch};
    constexpr int KB = 32 * BWD_NW;
    const dim3 gkv(la::cdiv(kv_len, KB) * n_head * batch), blk(64 * BWD_NW);
    hipLaunchKernelGGL(attention_x2_bwd_kv_kernel<1>, gkv, blk, KV_LDS, stream, p);
    hipLaunchKernelGGL(attention_x2_bwd_kv_kernel<2>, gkv, blk, KV_LDS, stream, p);
    hipLaunchKernelGGL(attention_x2_bwd_q_kernel, dim3(la::cdiv(q_len, KB) * n_head * batch), dim3(64 * BWD_NW), Q_LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
