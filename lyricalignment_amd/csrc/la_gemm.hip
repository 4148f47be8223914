// la_gemm.hip -- C = epi(A W^T): every nn.Linear / nn.Conv1d-as-GEMM of the encoder and
// the GRU input projections (whisper.model.AudioEncoder; module/align_model.py:23-33).
// Main loop: la_gemm_core.h.  bf16 operands use v_mfma_f32_16x16x32_bf16, f32 operands
// v_mfma_f32_16x16x4_f32 (exact fmaf chains: the parity mode).
#include <type_traits>

#include "la_gemm_core.h"

using la::bf16_t;
using namespace la::gemm;

namespace {

struct GemmParams {
    int M, N, K;
    const void *A;
    int64_t lda, strideA;
    const void *W;
    int64_t strideW;
    void *C;
    int64_t ldc, strideC;
    const float *bias;
    int64_t strideBias;
    const float *residual;
    int64_t ldr, strideR;
    int epilogue;
    int tiles_m, tiles_n;
};

template <typename T, bool OUT_F32>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const T *A = reinterpret_cast<const T *>(p.A) + (int64_t)z * p.strideA;
    const T *W = reinterpret_cast<const T *>(p.W) + (int64_t)z * p.strideW;
    const float *bias = p.bias ? p.bias + (int64_t)z * p.strideBias : nullptr;

    f32x4 acc[4][4];
    mainloop<T>(A, p.lda, p.M, W, p.K, p.N, p.K, m0, n0, lds, acc);

    typedef typename std::conditional<OUT_F32, float, T>::type TC;
    TC *C = reinterpret_cast<TC *>(p.C) + (int64_t)z * p.strideC;
    const float *R = p.residual ? p.residual + (int64_t)z * p.strideR : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const bool vec_c = (p.ldc % 4 == 0) && (p.N % 4 == 0);
    const bool vec_r = R && (p.ldr % 4 == 0) && (p.N % 4 == 0);
    const bool has_bias = (p.epilogue & LA_EPI_BIAS) && bias;
    const bool do_gelu = p.epilogue & LA_EPI_GELU;
    const bool do_mish = p.epilogue & LA_EPI_MISH;
    const bool do_res = (p.epilogue & LA_EPI_RESIDUAL) && R;

#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + q * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
            const int nv = min(4, p.N - n);
            if (has_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) v[j] += bias[n + j];
            }
            if (do_gelu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = la::gelu_erf(v[j]);
            }
            if (do_mish) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = la::mish(v[j]);
            }
            if (do_res) {
                const float *rr = R + (int64_t)m * p.ldr + n;
                if (vec_r) {
                    const float4 t = *reinterpret_cast<const float4 *>(rr);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < nv) v[j] += rr[j];
                }
            }
            TC *c = C + (int64_t)m * p.ldc + n;
            if (vec_c) {
                if constexpr (sizeof(TC) == 4) {
                    *reinterpret_cast<float4 *>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    ushort4 pk;
                    pk.x = la::f32_to_bf16(v[0]); pk.y = la::f32_to_bf16(v[1]);
                    pk.z = la::f32_to_bf16(v[2]); pk.w = la::f32_to_bf16(v[3]);
                    *reinterpret_cast<ushort4 *>(c) = pk;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) la::Elem<TC>::store(c + j, v[j]);
            }
        }
    }
}

template <typename T, bool OUT_F32>
int launch(const GemmParams &p, int batch, hipStream_t stream, const char *family) {
    auto kern = gemm_kernel<T, OUT_F32>;
    static bool attr_done = false;
    if (!attr_done) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    la::TimerScope ts(family, stream);
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, batch), dim3(NTHREADS), LDS_BYTES, stream, p);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

}  // namespace

int la::gemm_run(int dtype, int M, int N, int K, int batch, const void *A, int64_t lda, int64_t strideA, const void *W,
                 int64_t strideW, void *C, int64_t ldc, int64_t strideC, const float *bias, int64_t strideBias,
                 const float *residual, int64_t ldr, int64_t strideR, int epilogue, hipStream_t stream) {
    if (M == 0 || N == 0 || batch == 0) return LA_OK;
    LA_CHECK_ARG(A && W && C, "gemm: null pointer");
    LA_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm: bad sizes");
    LA_CHECK_ARG(dtype == LA_F32 || dtype == LA_BF16, "gemm: bad dtype");
    const int ke = dtype == LA_BF16 ? 64 : 32;
    const int es = dtype == LA_BF16 ? 2 : 4;
    LA_CHECK_ARG(K % ke == 0, "gemm: K=%d must be a multiple of %d", K, ke);
    LA_CHECK_ARG((lda * es) % 16 == 0 && (strideA * es) % 16 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0),
                 "gemm: A/W rows must be 16-byte aligned");
    LA_CHECK_ARG(!(epilogue & LA_EPI_RESIDUAL) || residual, "gemm: residual epilogue without pointer");
    LA_CHECK_ARG(!(epilogue & LA_EPI_BIAS) || bias, "gemm: bias epilogue without pointer");
    LA_CHECK_ARG((strideW * es) % 16 == 0, "gemm: W batch stride must be 16-byte aligned");
    GemmParams p{M, N, K, A, lda, strideA, W, strideW, C, ldc, strideC, bias, strideBias, residual, ldr, strideR, epilogue,
                 la::cdiv(M, BM), la::cdiv(N, BN)};
    const bool out_f32 = epilogue & LA_EPI_OUT_F32;
    if (dtype == LA_BF16)
        return out_f32 ? launch<bf16_t, true>(p, batch, stream, "gemm_bf16") : launch<bf16_t, false>(p, batch, stream, "gemm_bf16");
    return launch<float, true>(p, batch, stream, "gemm_f32");
}

extern "C" int la_gemm(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda,
                       int64_t strideA, const void *W, void *C, int64_t ldc, int64_t strideC, const float *bias,
                       const float *residual, int64_t ldr, int64_t strideR, int32_t epilogue, void *stream_) {
    return la::gemm_run(dtype, M, N, K, batch, A, lda, strideA, W, 0, C, ldc, strideC, bias, 0, residual, ldr, strideR,
                        epilogue, (hipStream_t)stream_);
}
