// la_attention_f16x2.hip -- the float32 attention of the fine-tune step on the f16 matrix pipe at float32 accuracy ("f16x2", la_f32x2.hip /
// la_x2.h: a float32 value scaled by a power of two is hi + lo in IEEE half, a product a b = a_lo b_hi + a_hi b_lo + a_hi b_hi in float32
// accumulate).  The reference trains in float32 (train_multitask.py:325-326 through whisper/model.py MultiHeadAttention.qkv_attention);
// gfx950 multiplies float32 operands at 1/16 of its 16-bit rate, and after the Linear layers moved to the f16x2 GEMM the float32-MFMA
// attention kernels (la_attention.hip attention_f32_kernel, la_attention_bwd.hip) were a third of the optimizer step.
//
// Forward (this file's first half): the register-resident flash form of the 16-bit inference kernel (la_attention.hip: the score tile is
// computed transposed, S^T = K Q^T, so a lane holds one query's scores; the exponentiated tile is directly the B operand of
// O^T += V^T P^T; K and V tiles staged by LDS-DMA into swizzled images, V^T fragments by ds_read_b64_tr_b16) with every product as three
// 32x32x16 f16 MFMAs.  An earlier attempt built on the 64 x 64 LDS-tiled float32 kernels (lab/la_attention_x2.hip) was bound by LDS round
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
__global__ __launch_bounds__(1024) void heads_split_kernel(const float *x, int64_t ld, int T, int H, unsigned short *planes, float *inv_scale) {
    __shared__ float red[16];
    const int head = blockIdx.x, clip = blockIdx.y, tid = threadIdx.x;
    const int c4 = tid & 15, r0 = tid >> 4;
    const float *src = x + (int64_t)clip * T * ld + head * 64 + c4 * 4;
    float mx = 0.f;
    for (int r = r0; r < T; r += 64) {
        const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)r * ld);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
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
    if (tid == 0) inv_scale[clip * H + head] = inv;
    const int64_t C = 64 * (int64_t)H;
    unsigned short *dst = planes + (int64_t)clip * T * 2 * C + head * 64 + c4 * 4;
    for (int r = r0; r < T; r += 64) {
        const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)r * ld);
        const unsigned a = la::x2::pack_hi_lo(v.x * s), b = la::x2::pack_hi_lo(v.y * s), c = la::x2::pack_hi_lo(v.z * s), d = la::x2::pack_hi_lo(v.w * s);
        *reinterpret_cast<uint2 *>(dst + (int64_t)r * 2 * C) = make_uint2((a & 0xffffu) | (b << 16), (c & 0xffffu) | (d << 16));
        *reinterpret_cast<uint2 *>(dst + (int64_t)r * 2 * C + C) = make_uint2((a >> 16) | (b & 0xffff0000u), (c >> 16) | (d & 0xffff0000u));
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
    // (v_exp_f32) and one between the two partial writes of the destination, as hipcc places them around its own v_fma_mix
    asm("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
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
        if (t + 1 < nkv) stage(t + 1, lds0 + (cur ^ 1) * 4 * IMG);
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
                    kfh[sub][c] = *reinterpret_cast<const uint4 *>(kl + off);
                    kfl[sub][c] = *reinterpret_cast<const uint4 *>(kl + IMG + off);
                }
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
                s[sub][r] = __builtin_amdgcn_exp2f(fmaf(s[sub][r], kScale, mneg));
                s[sub][r + 1] = __builtin_amdgcn_exp2f(fmaf(s[sub][r + 1], kScale, mneg));
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
                    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + a0)), h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + a1));
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + IMG + a0)), l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vl + IMG + a1));
                    const f16x8 vh = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    const f16x8 vlo = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                    o[b] = mfma(vlo, phv, o[b]);
                    o[b] = mfma(vh, plv, o[b]);
                    o[b] = mfma(vh, phv, o[b]);
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
    hipLaunchKernelGGL(heads_split_kernel, dim3(n_head, batch), dim3(1024), 0, stream, q, ld_q, q_len, n_head, ws.qp, ws.sq);
    hipLaunchKernelGGL(heads_split_kernel, dim3(n_head, batch), dim3(1024), 0, stream, k, ld_kv, kv_len, n_head, ws.kp, ws.sk);
    hipLaunchKernelGGL(heads_split_kernel, dim3(n_head, batch), dim3(1024), 0, stream, v, ld_kv, kv_len, n_head, ws.vp, ws.sv);
    FwdParams p{ws.qp, ws.kp, ws.vp, ws.sq, ws.sk, ws.sv, out, ld_out, lse, q_len, kv_len, n_head, causal ? 1 : 0, batch};
    hipLaunchKernelGGL((attention_x2_fwd_kernel<8>), dim3(la::cdiv(q_len, 256) * n_head * batch), dim3(512), 0, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
