// la_gemm.hip -- C = epi(A W^T): every nn.Linear / nn.Conv1d-as-GEMM of the encoder and
// the GRU input projections (whisper.model.AudioEncoder; module/align_model.py:23-33).
// This file: the host dispatch (shape -> kernel), the 128 x 128 kernel family (la_gemm_core.h main loop: float32 operands with
// v_mfma_f32_16x16x4_f32 -- exact fmaf chains, the parity / training mode -- and the small 16-bit shapes), float32 split-K.
// The 256 x 256 kernel of the large 16-bit shapes is built in la_gemm_pp_bf16.hip / la_gemm_pp_f16.hip (la_gemm_pp_kernel.h).
#include <algorithm>
#include <type_traits>

#include "la_gemm_core.h"
#include "la_gemm_params.h"

using la::bf16_t;
using namespace la::gemm;

namespace {

template <typename T, bool OUT_F32, typename CF, bool TA = false, bool TW = false>
__global__ __launch_bounds__(CF::THREADS, 2) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const TileCoord tc = tile_coord(tile, p.tiles_m, p.tiles_n, p.group);
    const int m0 = tc.tm * CF::TM, n0 = tc.tn * BN;
    const int z = blockIdx.y;
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)z * p.strideA;
    const T *W = reinterpret_cast<const T *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;

    f32x4 acc[4][4];
    mainloop<T, CF, TA, TW>(A, p.lda, p.M, W, p.ldw, p.N, (p.K_tail && z == (int)gridDim.y - 1) ? p.K_tail : p.K, m0, n0, lds, acc);

    typedef typename std::conditional<OUT_F32, float, T>::type TC;
    TC *C = reinterpret_cast<TC *>(p.C) + (int64_t)z * p.strideC;
    const float *R = p.residual ? p.residual + (int64_t)z * p.strideR : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    const bool do_gelu = p.epilogue & LA_EPI_GELU;
    const bool do_mish = p.epilogue & LA_EPI_MISH;
    const bool do_res = (p.epilogue & LA_EPI_RESIDUAL) && R;

    // bias + activation in registers (lane: 4 consecutive columns n of row m); the activation switches are
    // wave-uniform and hoisted so each variant is a straight-line pass over the 64 accumulators
    if (has_bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + q * 4;
            float b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = bias[min(n + j, p.N - 1)];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mi][ni][j] += b4[j];
        }
    }
    if (do_gelu) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if constexpr (!OUT_F32 && sizeof(T) == 2) {      // result is rounded to 16 bits: the same form as the 256x256 kernel's
                    if (p.epilogue & 4096) {                     // (LA_GELU_PK=1: the erfc form on the packed pipe)
                        const la::f32x2 lo = la::gelu_pk(la::f32x2{acc[mi][ni][0], acc[mi][ni][1]});
                        const la::f32x2 hi = la::gelu_pk(la::f32x2{acc[mi][ni][2], acc[mi][ni][3]});
                        acc[mi][ni] = f32x4{lo.x, lo.y, hi.x, hi.y};
                    } else {
                        if (LA_DEV_BIT(p.epilogue, 8192)) {      // experiment build (LA_GELU_PK=2): one value at a time
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[mi][ni][j] = la::gelu_sig(acc[mi][ni][j]);
                        } else {
                            const la::f32x2 lo = la::gelu_sig2(la::f32x2{acc[mi][ni][0], acc[mi][ni][1]});
                            const la::f32x2 hi = la::gelu_sig2(la::f32x2{acc[mi][ni][2], acc[mi][ni][3]});
                            acc[mi][ni] = f32x4{lo.x, lo.y, hi.x, hi.y};
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mi][ni][j] = la::gelu_erf(acc[mi][ni][j]);
                }
            }
    } else if (do_mish) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mi][ni][j] = la::mish(acc[mi][ni][j]);
    }

    // Store path.  Each wave transposes its 64x64 tile through LDS (two passes of 32 rows, f32, row pitch 272 B:
    // conflict-free b128 writes and reads) so that 16 lanes cover 64 CONSECUTIVE columns of one row: full 128-B
    // (bf16) / 256-B (f32) line segments per row instead of 32-B pieces scattered over 16 rows (measured: the
    // scattered form ran the C write at 1.4 TB/s and cost 30-45 % of the kernel).
    constexpr int PITCH = 272;
    __syncthreads();  // every wave is done reading the operand stages
    unsigned char *reg = lds + wave * (32 * PITCH);
    const int wrow0 = m0 + wm * 64, wcol0 = n0 + wn * 64;
    const bool fast_c = ((p.ldc * (int64_t)sizeof(TC)) % 16 == 0) && ((uintptr_t)C % 16 == 0);
    const bool fast_r = do_res && (p.ldr % 4 == 0) && ((uintptr_t)R % 16 == 0);
    const bool probe_nostore = LA_DEV_BIT(p.epilogue, 256);  // experiment build: main loop without the C stores
    // interior wave tiles: straight-line form (see the ping-pong kernel's epilogue)
    if (wrow0 + 64 <= p.M && wcol0 + 64 <= p.N && fast_c && (!do_res || fast_r) && !probe_nostore) {
        TC *cw = C + (int64_t)wrow0 * p.ldc + wcol0 + r * 4;
        const float *rw = do_res ? R + (int64_t)wrow0 * p.ldr + wcol0 + r * 4 : nullptr;
        auto fast = [&](auto resc) {
            constexpr bool RES = decltype(resc)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 t[8];
                if constexpr (RES) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) t[it] = *reinterpret_cast<const float4 *>(rw + (int64_t)(h * 32 + it * 4 + q) * p.ldr);
                }
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int rl = it * 4 + q;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16);
                    if constexpr (RES) { v[0] += t[it].x; v[1] += t[it].y; v[2] += t[it].z; v[3] += t[it].w; }
                    TC *c = cw + (int64_t)(h * 32 + rl) * p.ldc;
                    if constexpr (sizeof(TC) == 4) {
                        *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        *reinterpret_cast<ushort4 *>(c) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        };
        if (do_res) fast(std::true_type{}); else fast(std::false_type{});
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4 *>(reg + (mm * 16 + r) * PITCH + (ni * 16 + q * 4) * 4) = acc[2 * h + mm][ni];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + q;                       // row inside this 32-row pass
            const int m = wrow0 + h * 32 + rl;
            const int n = wcol0 + r * 4;                     // 4 consecutive columns
            f32x4 v = *reinterpret_cast<const f32x4 *>(reg + rl * PITCH + r * 16);
            if (m >= p.M || n >= p.N) continue;
            if (probe_nostore && v[0] != 12345.678f) continue;
            const int nv = min(4, p.N - n);
            if (do_res) {
                const float *rr = R + (int64_t)m * p.ldr + n;
                if (fast_r && nv == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(rr);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                } else {
                    for (int j = 0; j < nv; ++j) v[j] += rr[j];
                }
            }
            TC *c = C + (int64_t)m * p.ldc + n;
            if (fast_c && nv == 4) {
                if constexpr (sizeof(TC) == 4) {
                    *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    *reinterpret_cast<ushort4 *>(c) = la::Pack4<TC>::run(v[0], v[1], v[2], v[3]);
                }
            } else {
                for (int j = 0; j < nv; ++j) la::Elem<TC>::store(c + j, v[j]);
            }
        }
    }
}

template <typename T, bool OUT_F32, typename CF, bool TA = false, bool TW = false>
int launch(GemmParams p, int batch, hipStream_t stream, const char *family) {
    auto kern = gemm_kernel<T, OUT_F32, CF, TA, TW>;
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS));
        attr_once.mark();
    }
    p.tiles_m = la::cdiv(p.M, CF::TM);
    if (const char *g = la::dev_env("LA_GELU_PK")) p.epilogue |= atoi(g) == 2 ? 8192 : 4096;
    la::TimerScope ts(family, stream, 2.0 * p.M * p.N * (p.K_tail ? (double)((batch - 1) * (int64_t)p.K + p.K_tail) : (double)p.K * batch));
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(CF::THREADS), CF::LDS, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// Measured on the config-2 shapes (tools/kbench.py, round 1): the 128x128 / 2-workgroups-per-CU configuration beats the
// 256x128 / 3-stage / 1-workgroup-per-CU one by 5-15 % on every shape, so it is the default; LA_GEMM_TILE=256 selects
// the big tile for A/B runs.
bool use_big_tile(int M, int N, int batch) {
    return la::opts().gemm_tile == 256 && (int64_t)la::cdiv(M, 256) * la::cdiv(N, BN) * batch >= 512;
}

}  // namespace

// ldw: row pitch of W in elements (0 = dense [N][K])
// Split-K tail: C[m][n] = epilogue(sum_s P[s][m][n]) in a fixed order (deterministic), same epilogue order as the GEMM kernels
// (bias, activation, residual).
__global__ void splitk_reduce_kernel(const float *P, int S, int M, int N, float *C, int64_t ldc, const float *bias,
                                     const float *residual, int64_t ldr, int epilogue) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += P[(int64_t)s * M * N + i];
    if ((epilogue & LA_EPI_BIAS) && bias) v += bias[n];
    if (epilogue & LA_EPI_GELU) v = la::gelu_erf(v);
    else if (epilogue & LA_EPI_MISH) v = la::mish(v);
    if ((epilogue & LA_EPI_RESIDUAL) && residual) v += residual[(int64_t)m * ldr + n];
    C[(int64_t)m * ldc + n] = v;
}

// float32 kernel by operand layout: NT (both K-contiguous), TT (both [K][rows]: weight gradients), NT/TW (input gradients)
static int launch_f32(const GemmParams &p, int batch, bool tA, bool tW, hipStream_t stream) {
    typedef Cfg<2, 2> Small;
    if (tA && tW) return launch<float, true, Small, true, true>(p, batch, stream, "gemm_f32");
    if (tW) return launch<float, true, Small, false, true>(p, batch, stream, "gemm_f32");
    if (tA) return launch<float, true, Small, true, false>(p, batch, stream, "gemm_f32");
    return launch<float, true, Small>(p, batch, stream, "gemm_f32");
}

struct LnFuse {
    void *C2; int64_t ldc2, strideC2;
    const float *stats, *csum;
    float *part;
};

// Row statistics from the producers' per-segment partials (Chan et al. combination of equal-sized groups):
// mean = avg(mean_k), M2 = sum(M2_k) + 64 sum((mean_k - mean)^2), rstd = 1 / sqrt(M2 / (64 slots) + eps).
__global__ void ln_stats_finalize_kernel(const float2 *part, int slots, int M, float eps, float2 *stats) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    float mean = 0.f;
    for (int k = 0; k < slots; ++k) mean += part[(int64_t)k * M + row].x;
    mean /= (float)slots;
    float m2 = 0.f;
    for (int k = 0; k < slots; ++k) {
        const float2 pk = part[(int64_t)k * M + row];
        const float d = pk.x - mean;
        m2 += pk.y + 64.0f * d * d;
    }
    stats[row] = make_float2(mean, 1.0f / sqrtf(m2 / (64.0f * (float)slots) + eps));
}

static int gemm_run_ldw(int dtype, int M, int N, int K, int batch, const void *A, int64_t lda, int64_t strideA, const void *W,
                        int64_t ldw_arg, int64_t strideW, void *C, int64_t ldc, int64_t strideC, const float *bias, int64_t strideBias,
                 const float *residual, int64_t ldr, int64_t strideR, int epilogue, hipStream_t stream, const LnFuse *ln = nullptr) {
    if (M == 0 || N == 0 || batch == 0) return LA_OK;
    LA_CHECK_ARG(A && W && C, "gemm: null pointer");
    LA_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm: bad sizes");
    LA_CHECK_ARG(dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16, "gemm: bad dtype");
    const int ke = dtype == LA_F32 ? 32 : 64;
    const int es = dtype == LA_F32 ? 4 : 2;
    const bool tA = epilogue & LA_GEMM_TRANS_A, tW = epilogue & LA_GEMM_TRANS_W;
    epilogue &= ~(LA_GEMM_TRANS_A | LA_GEMM_TRANS_W);
    if (tA || tW) {
        LA_CHECK_ARG(dtype == LA_F32, "gemm: transposed operands are float32 only");
        LA_CHECK_ARG((!tA || (M % 4 == 0 && lda >= M)) && (!tW || (N % 4 == 0 && (ldw_arg == 0 || ldw_arg >= N))),
                     "gemm: a transposed operand [K][rows] needs rows %% 4 == 0 and a pitch >= rows");
        LA_CHECK_ARG(!ln, "gemm: no LayerNorm fold with transposed operands");
    }
    LA_CHECK_ARG(K % ke == 0 || ((tA || tW) && (tA || lda >= la::round_up(K, ke)) && (tW || (ldw_arg >= la::round_up(K, ke)))),
                 "gemm: K=%d must be a multiple of %d (or: a transposed operand, and K-contiguous rows pitched to the rounded-up K)", K, ke);
    LA_CHECK_ARG((lda * es) % 16 == 0 && (strideA * es) % 16 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0),
                 "gemm: A/W rows must be 16-byte aligned");
    LA_CHECK_ARG(!(epilogue & LA_EPI_RESIDUAL) || residual, "gemm: residual epilogue without pointer");
    LA_CHECK_ARG(!(epilogue & LA_EPI_BIAS) || bias, "gemm: bias epilogue without pointer");
    LA_CHECK_ARG((strideW * es) % 16 == 0 && (ldw_arg * es) % 16 == 0, "gemm: W batch stride / row pitch must be 16-byte aligned");
    GemmParams p{M, N, K, A, lda, strideA, W, ldw_arg > 0 ? ldw_arg : (int64_t)(tW ? N : K), strideW, C, ldc, strideC, bias, strideBias, residual, ldr, strideR, epilogue,
                 0, la::cdiv(N, BN), pick_group(K, es, la::cdiv(N, BN))};
    const bool out_f32 = epilogue & LA_EPI_OUT_F32;
    typedef Cfg<2, 2> Small;
    typedef Cfg<4, 3> Big;
    if (dtype == LA_BF16 || dtype == LA_F16) {
        // 256x256 ping-pong kernel once it can fill the chip (>= 192 tiles); LA_GEMM_TILE=512 forces it, 128/256 forbid it
        // (a 256-column tile on N <= 128 -- the gathered-label logits, N = Lmax + 1 -- would compute mostly padding)
        const int forced = la::opts().gemm_tile;
        const bool pp = forced == 512 || (forced == 0 && N > 128 && (int64_t)la::cdiv(M, 256) * la::cdiv(N, 256) * batch >= 192);
        const bool half = dtype == LA_F16;
        if (ln) {
            if (!pp || (epilogue & LA_EPI_MISH)) {
                la::set_error("gemm_fused_ln: shape M=%d N=%d batch=%d does not run on the 256x256 kernel the fusion is built into", M, N, batch);
                return LA_EUNSUPPORTED;
            }
            LA_CHECK_ARG(!ln->C2 || (out_f32 && ln->ldc2 == ldc && ln->strideC2 == strideC && ldc % 4 == 0 && (uintptr_t)ln->C2 % 8 == 0),
                         "gemm_fused_ln: the 16-bit copy accompanies an f32 result and shares its row pitch / batch stride");
            LA_CHECK_ARG(!(ln->C2 && ln->stats), "gemm_fused_ln: a GEMM is the producer or the consumer of a folded LayerNorm, not both");
            LA_CHECK_ARG(!ln->stats || ln->csum, "gemm_fused_ln: row statistics without the column sums of the folded weights");
            LA_CHECK_ARG(!ln->csum || batch == 1, "gemm_fused_ln: the LayerNorm epilogue takes batch 1");
            LA_CHECK_ARG(!(ln->C2 && ln->csum), "gemm_fused_ln: a GEMM is the producer or the consumer of a folded LayerNorm, not both");
            LA_CHECK_ARG(!ln->part || (ln->C2 && N % 64 == 0 && batch == 1 && (uintptr_t)ln->part % 8 == 0),
                         "gemm_fused_ln: partial statistics go with the 16-bit copy, N % 64 == 0, batch 1");
            p.C2 = ln->C2; p.ldc2 = ln->ldc2; p.strideC2 = ln->strideC2; p.ln_stats = ln->stats; p.ln_csum = ln->csum; p.ln_part = ln->part;
        }
        if (pp && !(epilogue & LA_EPI_MISH)) {
            return half ? launch_pp_f16(p, batch, out_f32, stream) : launch_pp_bf16(p, batch, out_f32, stream);
        }
        if (!half && use_big_tile(M, N, batch))
            return out_f32 ? launch<bf16_t, true, Big>(p, batch, stream, "gemm_bf16") : launch<bf16_t, false, Big>(p, batch, stream, "gemm_bf16");
        if (half) return out_f32 ? launch<la::f16_t, true, Small>(p, batch, stream, "gemm_bf16") : launch<la::f16_t, false, Small>(p, batch, stream, "gemm_bf16");
        return out_f32 ? launch<bf16_t, true, Small>(p, batch, stream, "gemm_bf16") : launch<bf16_t, false, Small>(p, batch, stream, "gemm_bf16");
    }
    // float32 (training / parity mode) with few tiles and a long K -- the text decoder's 80-row GEMMs are 8 tiles on 256 CUs,
    // each a serial K = 1024..4096 walk: K is cut into S equal chunks that run as S batch slots of the same kernel into a
    // partial-sum buffer, and a second kernel adds them in a fixed order and applies the epilogue.
    if (batch == 1) {
        const int tiles = la::cdiv(M, Small::TM) * la::cdiv(N, BN);
        // S: chunks of whole 32-element k-steps, the last one may be shorter.  Few tiles -> fill the chip (<= 256 workgroups);
        // a partial last round of tiles (192 or 384 tiles on 256 CUs) -> S = 4 / 2 makes the rounds finer (768 quarter / half
        // tiles = 3 full rounds) when K is long enough to pay for the partial-sum pass.
        const int ksteps = la::cdiv(K, 32);
        // (two 128x128 workgroups fit a CU, so 512 slots would fill it twice over: LA_GEMM_SPLIT_SLOTS=512 measured no
        //  difference on the fused fine-tune step, 868 vs 868 ms; one workgroup per CU stays the limit)
        const int kSlots = la::dev_env("LA_GEMM_SPLIT_SLOTS") ? atoi(la::dev_env("LA_GEMM_SPLIT_SLOTS")) : 256;
        int S = 1;
        while (S < 16 && tiles * S * 2 <= kSlots && ksteps / (S * 2) >= 4) S *= 2;
        if (S == 1 && K >= 1024 && (int64_t)M * N <= ((int64_t)4 << 20)) {
            if (tiles * 4 % 256 == 0 && tiles % 256 != 0 && tiles < 256) S = 4;
            else if (tiles * 2 % 256 == 0 && tiles % 256 != 0 && tiles < 512) S = 2;
        }
        if (S > 1 && la::opts().gemm_splitk && !LA_DEV_BIT(epilogue, 256)) {
            float *part = static_cast<float *>(la::stream_scratch(stream, la::SCRATCH_SPLITK, (size_t)S * M * N * sizeof(float)));
            if (!part) { la::set_error("gemm: split-K scratch allocation failed"); return LA_EHIP; }
            const int Kc = la::cdiv(ksteps, S) * 32;                   // S - 1 chunks of Kc, the last one takes the rest
            if (K - (S - 1) * Kc <= 0) S = la::cdiv(K, Kc);
            GemmParams ps{M, N, Kc, A, lda, tA ? (int64_t)Kc * lda : (int64_t)Kc, W, p.ldw, tW ? (int64_t)Kc * p.ldw : (int64_t)Kc, part, (int64_t)N, (int64_t)M * N, nullptr, 0, nullptr, 0, 0,
                          LA_EPI_OUT_F32, 0, la::cdiv(N, BN), pick_group(Kc, es, la::cdiv(N, BN))};
            ps.K_tail = K - (S - 1) * Kc;
            const int rc = launch_f32(ps, S, tA, tW, stream);
            if (rc == LA_OK) {
                const int64_t total = (int64_t)M * N;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)la::cdiv(total, (int64_t)256)), dim3(256), 0, stream, part, S, M, N,
                                   reinterpret_cast<float *>(C), ldc, bias, residual, ldr, epilogue);
            }
            if (rc != LA_OK) return rc;
            LA_LAUNCH_CHECK();
            return LA_OK;
        }
    }
    return launch_f32(p, batch, tA, tW, stream);
}

// LayerNorm folded into the GEMMs on either side of it (encoder blocks, 16-bit modes, shapes that run on the 256x256 kernel):
//   producer  (C2 != null): besides the f32 result rows (residual stream) also stores them rounded to `dtype` into C2;
//   consumer  (ln_stats != null): A = those raw rows, W = gamma-folded weights, ln_stats [M][2] = (mean, rstd) per row,
//             ln_csum [N] = row sums of W; bias must already hold b + W beta.  LA_EUNSUPPORTED for other shapes / f32.
extern "C" int la_gemm_fused_ln(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda,
                                int64_t strideA, const void *W, void *C, int64_t ldc, int64_t strideC, const float *bias,
                                const float *residual, int64_t ldr, int64_t strideR, int32_t epilogue, void *C2, int64_t ldc2,
                                int64_t strideC2, const float *ln_stats, const float *ln_csum, float *ln_part, void *stream_) {
    if (dtype != LA_BF16 && dtype != LA_F16) {
        la::set_error("gemm_fused_ln: 16-bit compute dtypes only");
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(C2 || ln_csum, "gemm_fused_ln: neither a second output nor the consumer's column sums given");
    const LnFuse ln{C2, ldc2, strideC2, ln_stats, ln_csum, ln_part};
    return gemm_run_ldw(dtype, M, N, K, batch, A, lda, strideA, W, 0, 0, C, ldc, strideC, bias, 0, residual, ldr, strideR, epilogue,
                        (hipStream_t)stream_, &ln);
}

// The residual GEMMs of the 16-bit encoder on the SPLIT stream (SplitRes in la_common.h): (hi, lo) <- epi(A W^T) + residual, where
// the residual is an f32 array (the stem: conv2 + positional embedding writes the stream) or the stream itself (in place:
// x += out-proj / x += mlp).  hi [M][ld] `dtype` is at the same time the raw A operand of the next LayerNorm-folded GEMM.
extern "C" int la_gemm_split(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda, int64_t strideA,
                             const void *W, void *hi, void *lo, int64_t ld, int64_t stride, const float *bias, const float *residual,
                             int64_t ldr, int64_t strideR, int32_t epilogue, float *ln_part, void *stream_) {
    if (M == 0 || N == 0 || batch == 0) return LA_OK;
    if (dtype != LA_BF16 && dtype != LA_F16) {
        la::set_error("gemm_split: 16-bit compute dtypes only");
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG(A && W && hi && lo, "gemm_split: null pointer");
    LA_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm_split: bad sizes");
    if (N <= 128 || (int64_t)la::cdiv(M, 256) * la::cdiv(N, 256) * batch < 192 || K % 64 != 0) {
        la::set_error("gemm_split: shape M=%d N=%d K=%d batch=%d does not run on the 256x256 kernel the split epilogue is built into", M, N, K, batch);
        return LA_EUNSUPPORTED;
    }
    LA_CHECK_ARG((epilogue & ~(LA_EPI_BIAS | LA_EPI_GELU | LA_EPI_RESIDUAL)) == 0, "gemm_split: epilogue takes BIAS, GELU, RESIDUAL only");
    LA_CHECK_ARG((lda * 2) % 16 == 0 && (strideA * 2) % 16 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0),
                 "gemm_split: A/W rows must be 16-byte aligned");
    LA_CHECK_ARG(!(epilogue & LA_EPI_BIAS) || bias, "gemm_split: bias epilogue without pointer");
    LA_CHECK_ARG(ld >= N && ((uintptr_t)hi % 2 == 0), "gemm_split: row pitch below N");
    LA_CHECK_ARG(!ln_part || (N % 64 == 0 && batch == 1 && (uintptr_t)ln_part % 8 == 0), "gemm_split: partial statistics need N %% 64 == 0, batch 1");
    {   // The stream is read and rewritten per tile while other tiles are still in their main loops: an operand that aliases it races.
        auto overlaps = [](const void *a, int64_t na, const void *b, int64_t nb) {
            const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
            return a0 < b0 + (uintptr_t)nb && b0 < a0 + (uintptr_t)na;
        };
        const int64_t a_bytes = ((int64_t)(batch - 1) * strideA + (int64_t)(M - 1) * lda + K) * 2, w_bytes = (int64_t)N * K * 2;
        const int64_t s_elems = (int64_t)(batch - 1) * stride + (int64_t)(M - 1) * ld + N;
        LA_CHECK_ARG(!overlaps(A, a_bytes, hi, s_elems * 2) && !overlaps(A, a_bytes, lo, s_elems) && !overlaps(W, w_bytes, hi, s_elems * 2) &&
                     !overlaps(W, w_bytes, lo, s_elems), "gemm_split: A / W must not overlap the (hi, lo) stream they update");
    }
    int epi = epilogue & (LA_EPI_BIAS | LA_EPI_GELU);
    if (epilogue & LA_EPI_RESIDUAL) epi |= residual ? LA_EPI_RESIDUAL : LA_EPI_SPLIT_INPLACE;
    GemmParams p{M, N, K, A, lda, strideA, W, (int64_t)K, 0, lo, ld, stride, bias, 0, residual, ldr, strideR, epi,
                 0, la::cdiv(N, BN), pick_group(K, 2, la::cdiv(N, BN))};
    p.C2 = hi; p.ldc2 = ld; p.strideC2 = stride; p.ln_part = ln_part;
    hipStream_t stream = (hipStream_t)stream_;
    return dtype == LA_F16 ? launch_split_f16(p, batch, stream) : launch_split_bf16(p, batch, stream);
}

extern "C" int la_ln_stats_finalize(const float *part, int32_t slots, int32_t M, float eps, float *stats, void *stream_) {
    if (M == 0) return LA_OK;
    LA_CHECK_ARG(part && stats && slots > 0 && M > 0, "ln_stats_finalize: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    la::TimerScope ts("layernorm", stream);
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3(la::cdiv(M, 256)), dim3(256), 0, stream, reinterpret_cast<const float2 *>(part), slots, M,
                       eps, reinterpret_cast<float2 *>(stats));
    LA_LAUNCH_CHECK();
    return LA_OK;
}

int la::gemm_run(int dtype, int M, int N, int K, int batch, const void *A, int64_t lda, int64_t strideA, const void *W,
                 int64_t strideW, void *C, int64_t ldc, int64_t strideC, const float *bias, int64_t strideBias,
                 const float *residual, int64_t ldr, int64_t strideR, int epilogue, hipStream_t stream) {
    return gemm_run_ldw(dtype, M, N, K, batch, A, lda, strideA, W, 0, strideW, C, ldc, strideC, bias, strideBias, residual, ldr,
                        strideR, epilogue, stream);
}

extern "C" int la_gemm(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda,
                       int64_t strideA, const void *W, void *C, int64_t ldc, int64_t strideC, const float *bias,
                       const float *residual, int64_t ldr, int64_t strideR, int32_t epilogue, void *stream_) {
    return la::gemm_run(dtype, M, N, K, batch, A, lda, strideA, W, 0, C, ldc, strideC, bias, 0, residual, ldr, strideR,
                        epilogue, (hipStream_t)stream_);
}

// General face used by the backward pass: W rows with their own pitch (ldw >= K) and per-batch strides for A, W, C, bias.
extern "C" int la_gemm_ex(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda,
                          int64_t strideA, const void *W, int64_t ldw, int64_t strideW, void *C, int64_t ldc, int64_t strideC,
                          const float *bias, int32_t epilogue, void *stream_) {
    LA_CHECK_ARG(ldw >= ((epilogue & LA_GEMM_TRANS_W) ? N : K), "gemm_ex: ldw smaller than a row of W");
    return gemm_run_ldw(dtype, M, N, K, batch, A, lda, strideA, W, ldw, strideW, C, ldc, strideC, bias, 0, nullptr, 0, 0, epilogue,
                        (hipStream_t)stream_);
}
