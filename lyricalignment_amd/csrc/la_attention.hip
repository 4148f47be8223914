// la_attention.hip -- non-causal multi-head self-attention for the Whisper encoder
// (whisper.model.MultiHeadAttention.qkv_attention; head_dim = 64 for every model size),
// flash-style: the [T x T] score matrix never leaves the CU.
//
// Work split: grid = (ceil(T/128) query tiles, heads, clips); 4 waves per workgroup, each wave
// owns 32 query rows and sweeps the clip's keys in tiles of 64.  K and V tiles are staged by
// global_load_lds (16 B per lane) into a double-buffered, XOR-swizzled LDS image.
//
// The score tile is computed TRANSPOSED (S^T = K Q^T, 32 keys x 32 queries per MFMA) so that a
// lane holds one query's scores in its accumulator registers: the row max / row sum are
// register-local plus one cross-half exchange, and the exponentiated tile is directly the
// B-operand of O^T += V^T P^T (cdna guide, "An accumulator tile as the next MFMA's operand").
//   bf16: v_mfma_f32_32x32x16_bf16; V^T fragments come from ds_read_b64_tr_b16 on the row-major
//         V image; P is rounded to bf16 for the second product (f32 accumulate, f32 softmax).
//   f32 : v_mfma_f32_32x32x2_f32 (exact fmaf chains): parity mode, P stays f32.
// Softmax runs in the exp2 domain (v_exp_f32): p = exp2(s*log2e - m*log2e).
#include <type_traits>

#include "la_common.h"

using la::bf16_t;

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// the two 16-bit operand types of the "throughput" kernels: same kernel text, different MFMA / conversion
template <typename T16> struct Half16;
template <> struct Half16<bf16_t> {
    typedef bf16x8 vec8;
    typedef __bf16 elem;
    static constexpr unsigned kOnes = 0x3F803F80u;   // two 1.0
    static constexpr float kPLimit = 0x1p30f;        // optimistic softmax: largest row partial sum (hence P) taken without a redo
    __device__ static __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Half16<la::f16_t> {
    typedef f16x8 vec8;
    typedef _Float16 elem;
    static constexpr unsigned kOnes = 0x3C003C00u;
    static constexpr float kPLimit = 0x1p13f;        // f16 tops out at 65504
    __device__ static __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int QT = 128;   // queries per workgroup (4 waves x 32)
constexpr int KT = 64;    // keys per tile
constexpr float kLog2e = 1.4426950408889634f;

struct AttnParams {
    const void *q, *k, *v;   // rows [clip*q_len + i][head*64 ..] / [clip*kv_len + j][head*64 ..]
    int64_t ld_q, ld_kv;
    void *out;
    int64_t ld_out;
    int q_len, kv_len, n_head, causal;
    int64_t q_bs, kv_bs, out_bs;   // rows from one clip to the next (== q_len / kv_len / q_len unless the caller says otherwise)
    int batch;
    float defer_thr;               // 16-bit kernels: deferred-maximum threshold (exp2 domain); 0 = the textbook online softmax
    float *lse = nullptr;          // float32 kernel, training forward: log sum_j exp(s_ij) per query row, [batch][n_head][q_len] (la_attention_bwd_f32)
    float score_scale = 1.4426950408889634f;   // 16-bit kernels: log2(e), or 1 when q already carries it (LA_Q_LOG2)
};

// Block -> (query tile, head, clip).  Workgroups are dealt round-robin over the 8 XCDs in launch order and every XCD has its
// own 4 MiB L2, so with the natural order the query tiles of one (clip, head) -- which all sweep the same 384 KiB of K and V
// -- land on 8 different L2s and the chip-wide in-flight K/V set (~40 MB) thrashes every one of them.  With (clip, head)
// pairs a multiple of 8, launch slot L goes to XCD L % 8 and that XCD's slots walk (pair, query tile) with the tile
// fastest: a pair's tiles share one L2 and ~13 pairs are in flight per XCD.  Otherwise the natural order is kept.
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

__device__ __forceinline__ void glds16(const void *g, void *l) {
    la::glds16(g, l);
}

// ---------------------------------------------------------------------------------------------
// bf16
// ---------------------------------------------------------------------------------------------
// K image: [64 keys][128 B], 16-B slot s of row r stored at slot s ^ ((r >> 1) & 7)   (ds_read_b128 rows)
// V image: [64 keys][128 B], slot s of row r stored at slot s ^ (((r >> 1) & 1) << 2) (ds_read_b64_tr_b16)
__device__ __forceinline__ int kswz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int vswz(int r) { return ((r >> 1) & 1) << 2; }

// K/V staging: 2 pieces (8 rows x 128 B) of each per wave and tile.  The per-lane byte offset of a piece (tile row *
// pitch + source-side swizzled slot) is loop invariant; the tile advance is a scalar base bump.  Rows past the end of
// the clip (last tile only) are clamped to the last key -- they are masked to -inf in the scores.
struct KvOff { unsigned k[2], v[2]; };
template <int PER>     // pieces of K (and of V) per wave and tile: 8 / waves per workgroup
__device__ __forceinline__ KvOff kv_offsets_bf16(int64_t ld, int key0, int T, int wave, int lane) {
    KvOff o;
    const int r8 = lane >> 3, ps = lane & 7;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int r = (wave * PER + i) * 8 + r8;
        int rr = key0 + r < T ? r : T - 1 - key0;
        o.k[i] = (unsigned)(rr * ld * 2) + ((ps ^ kswz(r)) << 4);
        o.v[i] = (unsigned)(rr * ld * 2) + ((ps ^ vswz(r)) << 4);
    }
    return o;
}
template <int PER>
__device__ __forceinline__ void stage_kv_bf16(const bf16_t *kbase, const bf16_t *vbase, int64_t ld, int key0, const KvOff &o,
                                              unsigned kl, unsigned vl, int wave) {
    const bf16_t *ks = kbase + (int64_t)key0 * ld, *vs = vbase + (int64_t)key0 * ld;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        la::glds16_so(o.k[i], ks, kl + (wave * PER + i) * 1024);
        la::glds16_so(o.v[i], vs, vl + (wave * PER + i) * 1024);
    }
}

// NW = waves per workgroup (32 queries each): 4 (128 queries, 5 workgroups = 20 waves per CU by LDS) or 8 (256 queries: every
// K / V tile is staged once for twice the queries, 4 workgroups = 32 waves per CU)
// MSUM: the softmax denominator on the matrix pipe -- one extra MFMA per 16 keys with an all-ones A operand accumulates
// sum_k P[k][q] (of the ROUNDED P, like the numerator) in a third accumulator, instead of 32 v_add_f32 per key tile on the
// vector pipe, which is the busier one in this kernel (VALU 71 % / MFMA 45 % of the SIMD cycles).
// KO (diagnostic builds only, -DLA_ATTN_KNOCKOUT, results are garbage): a bit mask of the parts left out, to see which of the three
// pipes the tile time follows -- 1 no exponentials, 2 no V^T fragment reads, 4 no K fragment reads, 8 no staging after tile 0,
// 16 no MFMAs.
// OPT: optimistic softmax -- no per-tile maximum.  P = exp2(s log2 e - m_run) is formed against the running maximum as it
// stands and the row sums (needed anyway) are tested: only if some lane's partial sum is not <= LIMIT (a score rose far above
// the running maximum, or the first tile, where m_run = -inf gives +inf / NaN) is the tile redone the textbook way -- scores
// recomputed from the K tile still in LDS, maximum, rescale.  LIMIT keeps P inside the 16-bit type (2^30 for bf16, 2^13 for
// f16); the f32 accumulators and the final O / l are exact in the scale.
// PRIO (diagnostic builds): 1 = the S MFMA phase at raised wave priority, 2 = S and PV raised (softmax at 0), 3 = softmax raised
// FOLD (bf16, LA_Q_LOG2: q carries head_dim^-0.5 log2 e, so the scores arrive in the exp2 domain): the optimistic form with the
// running maximum STARTING AT 0 and, while no query of the wave has needed one, no subtraction at all -- P = exp2(s) as the scores
// come out of the matrix pipe.  bf16 has float32's exponent range, so P in 2^-126 .. 2^30 is as precise relative to its row as
// exp2(s - max) is; the float32 accumulators and the final O / l do not care about the common scale.  Per score that leaves
// v_exp_f32 + v_add_f32 (row sum) + half a v_cvt_pk on the vector pipe, the busier one here, instead of v_fma_f32 + those
// (337 -> 315 us per Whisper-medium layer).  Scores above 2^30 take the same redo path as before (it installs a maximum; from
// then on the wave subtracts it); a query whose scores ALL lie below ~2^-100 would underflow its sum: the block then repeats its
// sweep the textbook way (block-uniform decision after the sweep; never seen on real activations, forced in the tests).
template <typename T16, int NW = 4, bool MSUM = false, int KO = 0, bool OPT = false, int PRIO = 0, bool FOLD = false>
__global__ __launch_bounds__(64 * NW, FOLD ? 4 : 2) void attention_bf16_kernel(AttnParams p) {
    static_assert(!(OPT && MSUM), "the optimistic form tests the vector-pipe row sums");
    static_assert(!FOLD || OPT, "the log2-domain form is a variant of the optimistic one");
    constexpr int QT = 32 * NW, PER = 8 / NW;
    typedef Half16<T16> HT;
    typedef typename HT::vec8 vec8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * KT * 128];  // [buf][K|V][64][128 B] = 32 KiB
    const int T = p.kv_len;
    const BlockCoord bc = block_coord((p.q_len + QT - 1) / QT, p.n_head, p.batch);
    const int qt = bc.qt, head = bc.head, clip = bc.clip;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    // (pointer arithmetic in 16-bit elements; bf16_t stands for either storage type here)
    const bf16_t *base = reinterpret_cast<const bf16_t *>(p.q) + (int64_t)clip * p.q_bs * p.ld_q + head * 64;
    const bf16_t *kbase = reinterpret_cast<const bf16_t *>(p.k) + (int64_t)clip * p.kv_bs * p.ld_kv + head * 64;
    const bf16_t *vbase = reinterpret_cast<const bf16_t *>(p.v) + (int64_t)clip * p.kv_bs * p.ld_kv + head * 64;

    // Q fragments (B operand): lane (q = i32, h) holds Q[q][16c + 8h .. +8], c = 0..3
    int qrow = qt * QT + wave * 32 + i32;
    const bool q_valid = qrow < p.q_len;
    qrow = q_valid ? qrow : p.q_len - 1;
    uint4 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const uint4 *>(base + (int64_t)qrow * p.ld_q + 16 * c + 8 * h);

    f32x16 o[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_run = FOLD ? 0.f : -INFINITY, l_part = 0.f;
    int plain = FOLD ? 1 : 0;               // FOLD: no query of this wave has needed a maximum yet: P = exp2(s) as it stands (wave-uniform)
    const float kScale = FOLD ? 1.0f : p.score_scale;
    f32x16 osum;
#pragma unroll
    for (int r = 0; r < 16; ++r) osum[r] = 0.f;
    const uint4 ones4 = uint4{HT::kOnes, HT::kOnes, HT::kOnes, HT::kOnes};

    int nkv = (T + KT - 1) / KT;
    const int nkv_all = nkv;
    if (p.causal) nkv = min(nkv, (min(p.q_len, (qt + 1) * QT) - 1) / KT + 1);   // tiles above the block's diagonal are all masked
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(la::lds_addr_u32(lds));
    const KvOff off_full = kv_offsets_bf16<PER>(p.ld_kv, 0, KT, wave, lane);                      // every row valid
    const KvOff off_last = kv_offsets_bf16<PER>(p.ld_kv, (nkv_all - 1) * KT, T, wave, lane);      // rows clamped to key T-1
    stage_kv_bf16<PER>(kbase, vbase, p.ld_kv, 0, nkv_all == 1 ? off_last : off_full, lds0, lds0 + KT * 128, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // The Q fragments are "used" here so that hipcc places its own wait for their loads before the loop.  Left to itself it
    // waits at their first use inside the loop (vmcnt(3) .. vmcnt(0) ahead of the S MFMAs), and in steady state those counts
    // no longer refer to the Q loads but to the four staging loads of the NEXT tile just issued: every tile then stalls on
    // its successor's K / V arriving instead of letting them land under the softmax and PV phases.
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 t = __builtin_bit_cast(u32x4, qf[c]);
        asm volatile("" : "+v"(t));
        qf[c] = __builtin_bit_cast(uint4, t);
    }
    __syncthreads();

    // per-lane constant parts of the V^T transposed-read address
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;

    // the tile body is instantiated once per LDS buffer so every fragment address is (loop-invariant VGPR) + immediate
    auto tile = [&](int t, auto curc) __attribute__((always_inline)) {
        constexpr int cur = decltype(curc)::value;
        const unsigned char *kl = lds + cur * (2 * KT * 128);
        const unsigned char *vl = kl + KT * 128;
        if (t + 1 < nkv && !(KO & 8)) {
            const unsigned nk = lds0 + (cur ^ 1) * (2 * KT * 128);
            stage_kv_bf16<PER>(kbase, vbase, p.ld_kv, (t + 1) * KT, t + 2 == nkv_all ? off_last : off_full, nk, nk + KT * 128, wave);
        }
        // ---- S^T = K Q^T : two 32-key sub-tiles ----
        // all eight K fragments are requested before the first MFMA (32 VGPRs; the kernel has room): the reads return under
        // the MFMAs instead of one ds_read -> wait -> MFMA round trip per fragment
        f32x16 s[2];
        typedef std::integral_constant<int, 0> Sub0;
        typedef std::integral_constant<int, 1> Sub1;
        typedef std::integral_constant<int, 2> SubBoth;
        // scores of sub-tile `which` (2 = both).  serial: the redo path -- one fragment pair in flight (few live registers)
        auto scores = [&](auto serialc, auto whichc) {
            constexpr bool serial = decltype(serialc)::value;
            constexpr int lo = decltype(whichc)::value == 2 ? 0 : decltype(whichc)::value;
            constexpr int hi = decltype(whichc)::value == 2 ? 2 : lo + 1;
            uint4 kf[2][4];
#pragma unroll
            for (int sub = lo; sub < hi; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
            if constexpr (!serial) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int sub = lo; sub < hi; ++sub) {
                        const int row = sub * 32 + i32;
                        if constexpr (KO & 4) kf[sub][c] = qf[(c + sub) & 3];
                        else kf[sub][c] = *reinterpret_cast<const uint4 *>(kl + row * 128 + (((2 * c + h) ^ kswz(row)) << 4));
                    }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (serial) {
#pragma unroll
                    for (int sub = lo; sub < hi; ++sub) {
                        const int row = sub * 32 + i32;
                        kf[sub][c] = *reinterpret_cast<const uint4 *>(kl + row * 128 + (((2 * c + h) ^ kswz(row)) << 4));
                    }
                }
#pragma unroll
                for (int sub = lo; sub < hi; ++sub)
                    if constexpr (KO & 16) s[sub][c] += __builtin_bit_cast(float, kf[sub][c].x) + __builtin_bit_cast(float, kf[sub][c].w);
                    else s[sub] = HT::mfma32(__builtin_bit_cast(vec8, kf[sub][c]), __builtin_bit_cast(vec8, qf[c]), s[sub]);
                if constexpr (serial) __builtin_amdgcn_sched_barrier(0);
            }
            // ---- mask keys >= kv_len (last tile) and, for the causal decoder self-attention, keys after the query ----
            if ((t + 1) * KT > T || p.causal) {
                const int kmax = p.causal ? min(T - 1, qrow) : T - 1;
#pragma unroll
                for (int sub = lo; sub < hi; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + sub * 32 + acc_row(r, h) > kmax) s[sub][r] = -INFINITY;
            }
        };
        // s <- exp2(s log2 e - m), returns this lane's partial row sum
        // (two at a time: v_pk_fma_f32 / v_pk_add_f32, 16 + 16 instructions instead of 32 + 32)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto exponentials = [&](float m, auto whichc) -> float {
            constexpr int lo = decltype(whichc)::value == 2 ? 0 : decltype(whichc)::value;
            constexpr int hi = decltype(whichc)::value == 2 ? 2 : lo + 1;
            // plain v_fma_f32 / v_add_f32, two independent partial sums -- NOT packed f32: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32
            // do not run beside the matrix pipe at all (tools/valu_mfma_overlap.hip: 0 % of a block of them hides under MFMAs,
            // 45-70 % of any other vector instruction does) and cost 1.8x a plain instruction for 2x the work; measured here:
            // 344 us plain against 349 us packed
            float ps0 = 0.f, ps1 = 0.f;
            auto body = [&](auto subtract_m) {
#pragma unroll
                for (int sub = lo; sub < hi; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float a0, a1;
                        if constexpr (FOLD) {
                            a0 = s[sub][r]; a1 = s[sub][r + 1];
                            if constexpr (decltype(subtract_m)::value) { a0 -= m; a1 -= m; }
                        } else { a0 = fmaf(s[sub][r], kScale, -m); a1 = fmaf(s[sub][r + 1], kScale, -m); }
                        const float p0 = (KO & 1) ? a0 : __builtin_amdgcn_exp2f(a0), p1 = (KO & 1) ? a1 : __builtin_amdgcn_exp2f(a1);
                        s[sub][r] = p0;
                        s[sub][r + 1] = p1;
                        if constexpr (!MSUM) { ps0 += p0; ps1 += p1; }
                    }
            };
            if (FOLD && plain) body(std::false_type{});      // wave-uniform (an SGPR): only the redo path clears it
            else body(std::true_type{});
            return ps0 + ps1;
        };
        auto tile_max = [&](auto whichc) -> float {
            constexpr int lo = decltype(whichc)::value == 2 ? 0 : decltype(whichc)::value;
            constexpr int hi = decltype(whichc)::value == 2 ? 2 : lo + 1;
            float mx = -INFINITY;
#pragma unroll
            for (int sub = lo; sub < hi; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
            return fmaxf(mx, __shfl_xor(mx, 32)) * kScale;
        };
        auto rescale = [&](float alpha) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            if constexpr (MSUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) osum[r] *= alpha;
            }
        };
        // ---- O^T += V^T P^T for one 32-key sub-tile ----
        auto pv_sub = [&](auto subc) {
            constexpr int sub = decltype(subc)::value;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec8 pf;  // element j <-> accumulator register 8*ks + j <-> key sub*32 + 16ks + 8(j>>2) + 4h + (j&3)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (typename HT::elem)s[sub][8 * ks + j];
                if constexpr (MSUM) osum = HT::mfma32(__builtin_bit_cast(vec8, ones4), pf, osum);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // lane (dv = 32b + 16(g&1) + (lane&15), half h = g>>1): 4 keys key0 .. key0+3 per read
                    const int key0 = sub * 32 + 16 * ks + 4 * (g >> 1);
                    const int slot = b * 4 + 2 * (g & 1) + (pp >> 1);
                    const int r0 = key0 + q4, r1 = key0 + 8 + q4;
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 vf;
                    if constexpr (KO & 2) vf = __builtin_bit_cast(s16x8, qf[(2 * sub + ks + b) & 3]);
                    else {
                        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4 *)(vl + r0 * 128 + ((slot ^ vswz(r0)) << 4) + (pp & 1) * 8));
                        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4 *)(vl + r1 * 128 + ((slot ^ vswz(r1)) << 4) + (pp & 1) * 8));
                        vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                    if constexpr (KO & 16) o[b][2 * sub + ks] += (float)vf[0] + (float)vf[7] + (float)pf[0] + (float)pf[7];
                    else o[b] = HT::mfma32(__builtin_bit_cast(vec8, vf), pf, o[b]);
                }
            }
        };
        // the textbook step for sub-tile(s) `which`, scores recomputed from the K tile in LDS (the optimistic form's redo path)
        auto redo = [&](auto whichc) -> float {
            scores(std::true_type{}, whichc);
            const float m_new = fmaxf(m_run, tile_max(whichc));
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);    // exp2(-inf) = 0 on the first tile
            m_run = m_new;
            plain = 0;
            rescale(alpha);
            l_part *= alpha;
            return exponentials(m_new, whichc);
        };
        if constexpr (PRIO == 1 || PRIO == 2) __builtin_amdgcn_s_setprio(3);
        scores(std::false_type{}, SubBoth{});
        if constexpr (PRIO == 1 || PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(3);
        if constexpr (OPT) {
            // per 32-key sub-tile, so that sub-tile 1's exponentials and sub-tile 0's PV MFMAs share a basic block (hipcc
            // interleaves them, as it does in the textbook form)
            float ps0 = exponentials(m_run, Sub0{});
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ps0 <= HT::kPLimit)) != 0, 0)) ps0 = redo(Sub0{});   // wave-uniform, rare
            l_part += ps0;
            float ps1 = exponentials(m_run, Sub1{});
            pv_sub(Sub0{});
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ps1 <= HT::kPLimit)) != 0, 0)) ps1 = redo(Sub1{});
            l_part += ps1;
            if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(0);
            if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
            pv_sub(Sub1{});
            if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        } else {
            // ---- online softmax (exp2 domain) ----
            // Deferred maximum (cdna guide T13): the running maximum m_run only moves when some query's tile maximum exceeds it by
            // more than THR (in the exp2 domain); until then p = exp2(s - m_run) may reach 2^THR instead of 1 -- harmless in the
            // f32 accumulators and scale-free for the 16-bit rounding of P -- and the whole rescale of O (and of the running sum)
            // is skipped.  With scores of a few units the maximum of 1500 keys settles inside the first tile or two, so the
            // rescale block (32 multiplies per lane) runs on the first tiles only instead of on nearly every one.
            // thr <= 0: always move.
            const float mx2 = tile_max(SubBoth{});
            float m_new = m_run, alpha = 1.0f;
            if (__builtin_amdgcn_ballot_w64(mx2 > m_run + p.defer_thr) != 0) {     // wave-uniform
                m_new = fmaxf(m_run, mx2);
                alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // exp2(-inf) = 0 on the first tile
            }
            m_run = m_new;
            const float ps = exponentials(m_new, SubBoth{});
            if constexpr (!MSUM) l_part = l_part * alpha + ps;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) rescale(alpha);   // wave-uniform; exact: alpha == 1 changes nothing
            pv_sub(Sub0{});
            pv_sub(Sub1{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(KO & 32)) __syncthreads();       // (KO 32: the tile loop without its barrier -- timing only)
    };
    auto sweep = [&]() __attribute__((always_inline)) {
        for (int t = 0; t < nkv; t += 2) {
            tile(t, std::integral_constant<int, 0>{});
            if (t + 1 < nkv) tile(t + 1, std::integral_constant<int, 1>{});
        }
    };
    sweep();
    if constexpr (FOLD) {
        // a sum outside 2^-100 .. 2^100 (all of a query's scores far below 0, or an overflow the per-tile test did not see):
        // the whole block once more with a maximum from the first tile on (block-uniform decision, said so to the compiler;
        // a second, cold copy of the tile loop rather than a loop around the first: that nest cost 50 VGPRs)
        const float l0 = l_part + __shfl_xor(l_part, 32);
        if (__builtin_amdgcn_readfirstlane(__syncthreads_or(q_valid && !(l0 >= 0x1p-100f && l0 <= 0x1p100f)))) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
            m_run = -INFINITY; l_part = 0.f; plain = 0;
            stage_kv_bf16<PER>(kbase, vbase, p.ld_kv, 0, nkv_all == 1 ? off_last : off_full, lds0, lds0 + KT * 128, wave);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            sweep();
        }
    }

    // ---- epilogue: O[q][dv] = O^T / l ----
    const float l = MSUM ? osum[0] : l_part + __shfl_xor(l_part, 32);      // every row of osum holds the query's whole sum
    const float inv = 1.0f / l;
    if (q_valid) {
        bf16_t *orow = reinterpret_cast<bf16_t *>(p.out) + ((int64_t)clip * p.out_bs + qrow) * p.ld_out + head * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                *reinterpret_cast<ushort4 *>(orow + 32 * b + 8 * r4 + 4 * h) =
                    la::Pack4<T16>::run(o[b][4 * r4 + 0] * inv, o[b][4 * r4 + 1] * inv, o[b][4 * r4 + 2] * inv, o[b][4 * r4 + 3] * inv);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// f32 (parity mode)
// ---------------------------------------------------------------------------------------------
// K / V images: [64 keys][256 B], 16-B slot s of row r stored at slot s ^ (r & 15).
__device__ __forceinline__ void stage_kv_f32(const float *kbase, const float *vbase, int64_t ld, int key0, int T,
                                             unsigned char *kl, unsigned char *vl, int wave, int lane) {
    const int r4 = lane >> 4, ps = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        const int r = piece * 4 + r4;
        int key = key0 + r;
        key = key < T ? key : T - 1;
        const int ls = ps ^ (r & 15);
        glds16(kbase + (int64_t)key * ld + (ls << 2), kl + piece * 1024);
        glds16(vbase + (int64_t)key * ld + (ls << 2), vl + piece * 1024);
    }
}

__global__ __launch_bounds__(256, 2) void attention_f32_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [buf][K|V][64][256 B] = 64 KiB
    const int T = p.kv_len;
    const BlockCoord bc = block_coord((p.q_len + QT - 1) / QT, p.n_head, p.batch);
    const int qt = bc.qt, head = bc.head, clip = bc.clip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i32 = lane & 31, h = lane >> 5;
    const float *base = reinterpret_cast<const float *>(p.q) + (int64_t)clip * p.q_bs * p.ld_q + head * 64;
    const float *kbase = reinterpret_cast<const float *>(p.k) + (int64_t)clip * p.kv_bs * p.ld_kv + head * 64;
    const float *vbase = reinterpret_cast<const float *>(p.v) + (int64_t)clip * p.kv_bs * p.ld_kv + head * 64;

    int qrow = qt * QT + wave * 32 + i32;
    const bool q_valid = qrow < p.q_len;
    qrow = q_valid ? qrow : p.q_len - 1;
    // lane (q, h) holds Q[q][8c + 4h + e], c = 0..7, e = 0..3: element e feeds the e-th MFMA of chunk c
    float4 qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const float4 *>(base + (int64_t)qrow * p.ld_q + 8 * c + 4 * h);

    f32x16 o[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_run = -INFINITY, l_part = 0.f;

    constexpr int TILE = KT * 256;
    int nkv = (T + KT - 1) / KT;
    if (p.causal) nkv = min(nkv, (min(p.q_len, (qt + 1) * QT) - 1) / KT + 1);
    stage_kv_f32(kbase, vbase, p.ld_kv, 0, T, lds, lds + TILE, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < nkv; ++t) {
        const unsigned char *kl = lds + cur * (2 * TILE);
        const unsigned char *vl = kl + TILE;
        if (t + 1 < nkv) {
            unsigned char *nk = lds + (cur ^ 1) * (2 * TILE);
            stage_kv_f32(kbase, vbase, p.ld_kv, (t + 1) * KT, T, nk, nk + TILE, wave, lane);
        }
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
            const int row = sub * 32 + i32;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 kf = *reinterpret_cast<const float4 *>(kl + row * 256 + (((2 * c + h) ^ (row & 15)) << 4));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[c].x, s[sub], 0, 0, 0);
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[c].y, s[sub], 0, 0, 0);
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[c].z, s[sub], 0, 0, 0);
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[c].w, s[sub], 0, 0, 0);
            }
        }
        if ((t + 1) * KT > T || p.causal) {
            const int kmax = p.causal ? min(T - 1, qrow) : T - 1;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KT + sub * 32 + acc_row(r, h) > kmax) s[sub][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx * kLog2e);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(fmaf(s[sub][r], kLog2e, -m_new));
                s[sub][r] = pv;
                psum += pv;
            }
        l_part = l_part * alpha + psum;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
        // O^T += V^T P^T: MFMA `reg` contracts keys {acc_row(reg,0), acc_row(reg,1)}; B operand = s[sub][reg] as it stands,
        // A operand: lane (dv = 32b + i32, h) supplies V[key acc_row(reg,h)][dv]
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = sub * 32 + acc_row(r, h);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int col = 32 * b + i32;  // dv
                    const float vv = *reinterpret_cast<const float *>(vl + key * 256 + ((((col >> 2) ^ (key & 15)) << 4) | ((col & 3) << 2)));
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[sub][r], o[b], 0, 0, 0);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    const float l = l_part + __shfl_xor(l_part, 32);
    const float inv = 1.0f / l;
    if (p.lse && q_valid && h == 0)          // m_run is in the exp2 domain: lse = m ln 2 + ln l
        p.lse[((int64_t)clip * p.n_head + head) * p.q_len + qrow] = fmaf(m_run, 0.6931471805599453f, __logf(l));
    if (q_valid) {
        float *orow = reinterpret_cast<float *>(p.out) + ((int64_t)clip * p.out_bs + qrow) * p.ld_out + head * 64;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4 *>(orow + 32 * b + 8 * r4 + 4 * h) =
                    make_float4(o[b][4 * r4 + 0] * inv, o[b][4 * r4 + 1] * inv, o[b][4 * r4 + 2] * inv, o[b][4 * r4 + 3] * inv);
    }
}

}  // namespace

static int attention_launch(int dtype_arg, AttnParams p, int batch, hipStream_t stream) {
    const int dtype = dtype_arg & 0xff;
    const bool qlog2 = dtype_arg & LA_Q_LOG2;
    if (qlog2 && dtype != LA_BF16 && dtype != LA_F16) {
        la::set_error("attention: LA_Q_LOG2 goes with the 16-bit kernels");
        return LA_EINVAL;
    }
    p.score_scale = qlog2 ? 1.0f : kLog2e;
    p.batch = batch;
    const char *thr_env = la::dev_env("LA_ATTN_THR");         // experiment build only
    p.defer_thr = thr_env ? (float)atof(thr_env) : 8.0f;
    const dim3 grid(la::cdiv(p.q_len, QT) * p.n_head * batch), block(256);
    if (dtype == LA_BF16 || dtype == LA_F16) {
        la::TimerScope ts("attention_bf16", stream);
        // 256-query workgroups of 8 waves where a sequence has at least four of them (the encoder: T = 1500): every K / V tile is
        // staged once for twice the queries -- two LDS-DMA pieces per wave and tile instead of four, and the GEMM knock-outs of
        // round 4 price a piece at 60-185 cycles of issue -- same results, 325 against 333 us per layer alone, -0.25 ms per step
        // same-box (profiles/r4_ab_attention_8_waves.txt).  Option attn_nw = 4 | 8 (LA_ATTN_NW) forces either form.
        const int nw_opt = la::opts().attn_nw;
        const int nw = nw_opt ? nw_opt : (p.q_len >= 1024 ? 8 : 4);
        // Shipped forms: the optimistic-maximum tile loop (OPT) in both workgroup sizes; bfloat16 with exp2-domain scores (LA_Q_LOG2)
        // also folds the running maximum away until one is needed (FOLD).  The experiment build (-DLA_EXPERIMENTS) keeps the A/B
        // partners behind per-launch switches: LA_ATTN_OPT=0 (per-tile maximum, deferred by LA_ATTN_THR), LA_ATTN_FOLD=0,
        // LA_ATTN_MSUM (softmax denominator on the matrix pipe).
        const dim3 grid8(la::cdiv(p.q_len, 256) * p.n_head * batch), block8(512);
        const bool wide = nw == 8 && p.q_len >= 256;
#ifdef LA_EXPERIMENTS
        const char *opt_env = getenv("LA_ATTN_OPT"), *fold_env = getenv("LA_ATTN_FOLD");
        const bool no_opt = opt_env && atoi(opt_env) == 0, no_fold = fold_env && atoi(fold_env) == 0;
        if (wide && no_opt) {
            if (dtype == LA_F16) hipLaunchKernelGGL((attention_bf16_kernel<la::f16_t, 8>), grid8, block8, 0, stream, p);
            else hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 8>), grid8, block8, 0, stream, p);
        } else if (wide && dtype == LA_BF16 && qlog2 && no_fold) {
            hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 8, false, 0, true>), grid8, block8, 0, stream, p);
        } else if (!wide && getenv("LA_ATTN_MSUM")) {
            if (dtype == LA_F16) hipLaunchKernelGGL((attention_bf16_kernel<la::f16_t, 4, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4, true>), grid, block, 0, stream, p);
        } else if (!wide && no_opt) {
            if (dtype == LA_F16) hipLaunchKernelGGL((attention_bf16_kernel<la::f16_t, 4>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4>), grid, block, 0, stream, p);
        } else if (!wide && dtype == LA_BF16 && qlog2 && no_fold) {
            hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4, false, 0, true>), grid, block, 0, stream, p);
        } else
#endif
#ifdef LA_ATTN_KNOCKOUT
        if (const char *ko = getenv("LA_ATTN_KO"); ko && !wide) {  // diagnostic build: parts of the tile loop left out (bf16 only)
            switch (atoi(ko)) {
#define LA_KO_CASE(n) case n: hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4, false, n>), grid, block, 0, stream, p); break;
                LA_KO_CASE(1) LA_KO_CASE(2) LA_KO_CASE(4) LA_KO_CASE(6) LA_KO_CASE(8) LA_KO_CASE(14) LA_KO_CASE(16) LA_KO_CASE(17)
                LA_KO_CASE(7) LA_KO_CASE(15) LA_KO_CASE(30)
#undef LA_KO_CASE
            // LA_ATTN_LDSPAD=<bytes> of unused dynamic LDS per workgroup: 0 -> 4 workgroups per CU, 20480 -> 3, 49152 -> 2, 102400 -> 1
#define LA_KO_OPT(n, ko) case n: { const char *pad = getenv("LA_ATTN_LDSPAD"); \
        hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4, false, ko, true>), grid, block, pad ? atoi(pad) : 0, stream, p); } break;
                LA_KO_OPT(132, 32) LA_KO_OPT(140, 40) LA_KO_OPT(100, 0) LA_KO_OPT(101, 1) LA_KO_OPT(106, 6) LA_KO_OPT(108, 8) LA_KO_OPT(114, 14) LA_KO_OPT(116, 16) LA_KO_OPT(130, 30)
#undef LA_KO_OPT
#define LA_KO_PRIO(n, pr) case n: hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4, false, 0, true, pr>), grid, block, 0, stream, p); break;
                LA_KO_PRIO(201, 1) LA_KO_PRIO(202, 2) LA_KO_PRIO(203, 3)
#undef LA_KO_PRIO
                default: hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4>), grid, block, 0, stream, p);
            }
        } else
#endif
        if (wide) {
            if (dtype == LA_F16) hipLaunchKernelGGL((attention_bf16_kernel<la::f16_t, 8, false, 0, true>), grid8, block8, 0, stream, p);
            else if (qlog2) hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 8, false, 0, true, 0, true>), grid8, block8, 0, stream, p);
            else hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 8, false, 0, true>), grid8, block8, 0, stream, p);
        } else if (dtype == LA_F16) hipLaunchKernelGGL((attention_bf16_kernel<la::f16_t, 4, false, 0, true>), grid, block, 0, stream, p);
        else if (qlog2) hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4, false, 0, true, 0, true>), grid, block, 0, stream, p);   // exp2-domain scores, no maximum until one is needed
        else hipLaunchKernelGGL((attention_bf16_kernel<bf16_t, 4, false, 0, true>), grid, block, 0, stream, p);
    } else {
        static la::DeviceOnce attr_once;
        if (attr_once.pending()) {
            LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_f32_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KT * 256));
            attr_once.mark();
        }
        la::TimerScope ts("attention_f32", stream);
        hipLaunchKernelGGL(attention_f32_kernel, grid, block, 4 * KT * 256, stream, p);
    }
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_attention(int32_t dtype_arg, const void *qkv, int64_t ld_qkv, void *out, int64_t ld_out, int32_t batch,
                            int32_t frames, int32_t n_head, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || frames == 0) return LA_OK;
    LA_CHECK_ARG(qkv && out && batch > 0 && frames > 0 && n_head > 0, "attention: bad arguments");
    const int dtype = dtype_arg & 0xff;       // LA_Q_LOG2 may ride on it
    LA_CHECK_ARG((dtype_arg & ~(0xff | LA_Q_LOG2)) == 0 && (dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16), "attention: bad dtype");
    const int es = dtype == LA_F32 ? 4 : 2;
    LA_CHECK_ARG(ld_qkv >= 3 * n_head * 64 && ld_out >= n_head * 64, "attention: leading dimensions too small");
    LA_CHECK_ARG((ld_qkv * es) % 16 == 0 && (ld_out * es) % 16 == 0 && (uintptr_t)qkv % 16 == 0 && (uintptr_t)out % 16 == 0,
                 "attention: rows must be 16-byte aligned");
    const char *b = reinterpret_cast<const char *>(qkv);
    const int d = n_head * 64;
    AttnParams p{b, b + (int64_t)d * es, b + (int64_t)2 * d * es, ld_qkv, ld_qkv, out, ld_out, frames, frames, n_head, 0, frames, frames, frames, 0};
    return attention_launch(dtype_arg, p, batch, stream);
}

extern "C" int la_attention_ex(int32_t dtype_arg, const void *q, int64_t ld_q, const void *k, const void *v, int64_t ld_kv,
                               void *out, int64_t ld_out, int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head,
                               int32_t causal, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || q_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && out && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_ex: bad arguments");
    const int dtype = dtype_arg & 0xff;       // LA_Q_LOG2 may ride on it
    LA_CHECK_ARG((dtype_arg & ~(0xff | LA_Q_LOG2)) == 0 && (dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16), "attention_ex: bad dtype");
    LA_CHECK_ARG(!causal || q_len == kv_len, "attention_ex: causal masking is defined for self-attention (q_len == kv_len)");
    const int es = dtype == LA_F32 ? 4 : 2;
    LA_CHECK_ARG(ld_q >= n_head * 64 && ld_kv >= n_head * 64 && ld_out >= n_head * 64, "attention_ex: leading dimensions too small");
    LA_CHECK_ARG((ld_q * es) % 16 == 0 && (ld_kv * es) % 16 == 0 && (ld_out * es) % 16 == 0 && (uintptr_t)q % 16 == 0 &&
                     (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)out % 16 == 0,
                 "attention_ex: rows must be 16-byte aligned");
    AttnParams p{q, k, v, ld_q, ld_kv, out, ld_out, q_len, kv_len, n_head, causal ? 1 : 0, q_len, kv_len, q_len, 0};
    return attention_launch(dtype_arg, p, batch, stream);
}

// The float32 forward of the TRAINING path: la_attention_ex that also hands over the row statistic the fused backward needs
// (lse [batch][n_head][q_len]), so that la_attention_bwd_f32 does not recompute the scores once more just for it.
extern "C" int la_attention_lse_f32(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, float *out, int64_t ld_out,
                                    int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, float *lse, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || q_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && out && lse && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_lse: bad arguments");
    LA_CHECK_ARG(!causal || q_len == kv_len, "attention_lse: causal masking is defined for self-attention (q_len == kv_len)");
    LA_CHECK_ARG(ld_q >= n_head * 64 && ld_kv >= n_head * 64 && ld_out >= n_head * 64, "attention_lse: leading dimensions too small");
    LA_CHECK_ARG(ld_q % 4 == 0 && ld_kv % 4 == 0 && ld_out % 4 == 0 && (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 &&
                     (uintptr_t)out % 16 == 0, "attention_lse: rows must be 16-byte aligned");
    AttnParams p{q, k, v, ld_q, ld_kv, out, ld_out, q_len, kv_len, n_head, causal ? 1 : 0, q_len, kv_len, q_len, 0};
    p.lse = lse;
    return attention_launch(LA_F32, p, batch, stream);
}

// Attention against a key / value CACHE (autoregressive decoding, whisper/decoding.py's kv_cache): clip b's keys are rows
// [b * kv_batch_rows, b * kv_batch_rows + kv_len) of k / v -- the cache has room for kv_batch_rows >= kv_len rows per clip --
// and its queries rows [b * q_batch_rows, ... + q_len) of q; out rows are packed [b * q_len + i].  Queries are the LAST q_len
// positions of the sequence: with causal != 0 query i sees keys 0 .. kv_len - q_len + i.
extern "C" int la_attention_cached(int32_t dtype_arg, const void *q, int64_t ld_q, int64_t q_batch_rows, const void *k, const void *v,
                                   int64_t ld_kv, int64_t kv_batch_rows, void *out, int64_t ld_out, int32_t batch, int32_t q_len,
                                   int32_t kv_len, int32_t n_head, int32_t causal, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || q_len == 0) return LA_OK;
    LA_CHECK_ARG(q && k && v && out && batch > 0 && q_len > 0 && kv_len > 0 && n_head > 0, "attention_cached: bad arguments");
    const int dtype = dtype_arg & 0xff;       // LA_Q_LOG2 may ride on it
    LA_CHECK_ARG((dtype_arg & ~(0xff | LA_Q_LOG2)) == 0 && (dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16), "attention_cached: bad dtype");
    LA_CHECK_ARG(q_batch_rows >= q_len && kv_batch_rows >= kv_len, "attention_cached: batch strides shorter than the lengths");
    LA_CHECK_ARG(!causal || q_len == 1 || q_len == kv_len, "attention_cached: causal masking needs q_len == 1 or q_len == kv_len");
    const int es = dtype == LA_F32 ? 4 : 2;
    LA_CHECK_ARG(ld_q >= n_head * 64 && ld_kv >= n_head * 64 && ld_out >= n_head * 64, "attention_cached: leading dimensions too small");
    LA_CHECK_ARG((ld_q * es) % 16 == 0 && (ld_kv * es) % 16 == 0 && (ld_out * es) % 16 == 0 && (uintptr_t)q % 16 == 0 &&
                     (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)out % 16 == 0,
                 "attention_cached: rows must be 16-byte aligned");
    // one new token against the whole cache needs no mask at all
    AttnParams p{q, k, v, ld_q, ld_kv, out, ld_out, q_len, kv_len, n_head, (causal && q_len > 1) ? 1 : 0, q_batch_rows, kv_batch_rows, q_len, 0};
    return attention_launch(dtype_arg, p, batch, stream);
}
