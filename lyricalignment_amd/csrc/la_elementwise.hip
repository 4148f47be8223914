// la_elementwise.hip -- HBM-bound row kernels: emission prep from materialised
// logits, LayerNorm, casts, mel -> channels-last rows.  One wave64 (or one
// 256-thread workgroup) per row, 16-byte accesses wherever the row pitch allows.
#include "la_common.h"

using la::bf16_t;

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------
// emission prep from [B][T][V] f32 logits  (utils/alignment.py:123-134, :14-20)
// one 256-thread workgroup per (b, t) row; two passes over the row (max, then
// sum exp(x - max)) like torch's log_softmax; the second pass hits L2.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emissions_from_logits_kernel(
    const float *logits, int64_t bs, int64_t rs, int frames, int vocab, int variant, const int32_t *labels,
    int labels_stride, const int32_t *n_labels, int max_labels, float *em, int64_t em_bs, int64_t em_rs) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    const int b = row / frames, t = row % frames;
    const float *x = logits + (int64_t)b * bs + (int64_t)t * rs;
    const int c0 = variant == LA_VARIANT_CTC ? 1 : 0;
    const int c1 = variant == LA_VARIANT_CTC ? vocab - 1 : vocab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    float m = -INFINITY;
    for (int c = c0 + tid; c < c1; c += 256) m = fmaxf(m, x[c]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = c0 + tid; c < c1; c += 256) s += expf(x[c] - m);
    s = wave_sum(s);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    s = (red[4] + red[5]) + (red[6] + red[7]);
    const float logsum = logf(s);

    const int L = min(n_labels[b], max_labels);
    float *e = em + (int64_t)b * em_bs + (int64_t)t * em_rs;
    const int32_t *lab = labels + (int64_t)b * labels_stride;
    if (variant == LA_VARIANT_CTC) {
        const float xl = x[vocab - 1];
        const float sil = 1.0f / (1.0f + expf(-xl));       // F.sigmoid (:125)
        const float log_sil = logf(sil);                    // (:128)
        const float log_voiced = logf(1.0f - sil);          // naive form, may be -inf (:126,:129)
        if (tid == 0) e[0] = fmaxf(log_sil, -1000.0f);      // (:134)
        for (int n = tid; n < L; n += 256) {
            const int c = lab[n];
            float v = -1000.0f;
            if (c >= 1 && c < vocab - 1) v = fmaxf(((x[c] - m) - logsum) + log_voiced, -1000.0f);  // (:131-132)
            e[1 + n] = v;
        }
    } else {
        if (tid == 0) e[0] = fmaxf((x[0] - m) - logsum, -1000.0f);  // (:16,:20)
        for (int n = tid; n < L; n += 256) {
            const int c = lab[n];
            float v = -1000.0f;
            if (c >= 1 && c < vocab) v = fmaxf((x[c] - m) - logsum, -1000.0f);  // (:18)
            e[1 + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (d <= 64*4*MAXV)
// ---------------------------------------------------------------------------
template <typename TOut, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *x, int64_t ldx, int M, int d, const float *gamma,
                                                        const float *beta, TOut *y, int64_t ldy) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float *xr = x + (int64_t)row * ldx;
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            v[i] = *reinterpret_cast<const float4 *>(xr + c);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const float a = v[i].x - mean, b2 = v[i].y - mean, c2 = v[i].z - mean, d2 = v[i].w - mean;
            sq += (a * a + b2 * b2) + (c2 * c2 + d2 * d2);
        }
    }
    const float var = wave_sum(sq) / (float)d;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    TOut *yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
            const float4 bt = *reinterpret_cast<const float4 *>(beta + c);
            float o0 = (v[i].x - mean) * rstd * g.x + bt.x;
            float o1 = (v[i].y - mean) * rstd * g.y + bt.y;
            float o2 = (v[i].z - mean) * rstd * g.z + bt.z;
            float o3 = (v[i].w - mean) * rstd * g.w + bt.w;
            if constexpr (sizeof(TOut) == 4) {
                *reinterpret_cast<float4 *>(yr + c) = make_float4(o0, o1, o2, o3);
            } else {
                *reinterpret_cast<ushort4 *>(yr + c) = la::Pack4<TOut>::run(o0, o1, o2, o3);
            }
        }
    }
}

// The same LayerNorm over rows of the SPLIT residual stream (la_common.h SplitRes: hi 16-bit + lo byte per element), decoded on
// the way in; everything after the load is layernorm_kernel's arithmetic in the same order.
template <typename T16, typename TOut, int MAXV>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const unsigned short *hi, const unsigned char *lo, int64_t ldx, int M, int d,
                                                              const float *gamma, const float *beta, TOut *y, int64_t ldy) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const unsigned short *hr = hi + (int64_t)row * ldx;
    const unsigned char *lr = lo + (int64_t)row * ldx;
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const ushort4 h4 = *reinterpret_cast<const ushort4 *>(hr + c);
            const unsigned w = *reinterpret_cast<const unsigned *>(lr + c);
            v[i] = make_float4(la::split_decode<T16>(h4.x, (float)(w & 0xffu)), la::split_decode<T16>(h4.y, (float)((w >> 8) & 0xffu)),
                               la::split_decode<T16>(h4.z, (float)((w >> 16) & 0xffu)), la::split_decode<T16>(h4.w, (float)(w >> 24)));
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const float a = v[i].x - mean, b2 = v[i].y - mean, c2 = v[i].z - mean, d2 = v[i].w - mean;
            sq += (a * a + b2 * b2) + (c2 * c2 + d2 * d2);
        }
    }
    const float var = wave_sum(sq) / (float)d;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    TOut *yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
            const float4 bt = *reinterpret_cast<const float4 *>(beta + c);
            float o0 = (v[i].x - mean) * rstd * g.x + bt.x;
            float o1 = (v[i].y - mean) * rstd * g.y + bt.y;
            float o2 = (v[i].z - mean) * rstd * g.z + bt.z;
            float o3 = (v[i].w - mean) * rstd * g.w + bt.w;
            if constexpr (sizeof(TOut) == 4) {
                *reinterpret_cast<float4 *>(yr + c) = make_float4(o0, o1, o2, o3);
            } else {
                *reinterpret_cast<ushort4 *>(yr + c) = la::Pack4<TOut>::run(o0, o1, o2, o3);
            }
        }
    }
}

// Row statistics of 16-bit rows (the raw copy of the residual stream that the LayerNorm-folded GEMMs consume):
// stats[row] = (mean, 1 / sqrt(var + eps)), two-pass on the row held in registers.  One wave per row, d <= 64*8*MAXV.
template <typename T16, int MAXV>
__global__ __launch_bounds__(256) void row_stats16_kernel(const T16 *x, int64_t ldx, int M, int d, float eps, float2 *stats) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const T16 *xr = x + (int64_t)row * ldx;
    float v[MAXV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(xr + c);
            const T16 *e = reinterpret_cast<const T16 *>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = la::Elem<T16>::load(e + j); sum += v[i][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = v[i][j] - mean; sq = fmaf(a, a, sq); }
        }
    }
    const float var = wave_sum(sq) / (float)d;
    if (lane == 0) stats[row] = make_float2(mean, 1.0f / sqrtf(var + eps));
}

__global__ void cast_f32_bf16_kernel(const float *x, bf16_t *y, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        ushort4 pk;
        pk.x = la::f32_to_bf16(v.x); pk.y = la::f32_to_bf16(v.y);
        pk.z = la::f32_to_bf16(v.z); pk.w = la::f32_to_bf16(v.w);
        *reinterpret_cast<ushort4 *>(y + i) = pk;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t j = n & ~(int64_t)3; j < n; ++j) y[j] = la::f32_to_bf16(x[j]);
}

__global__ void cast_bf16_f32_kernel(const bf16_t *x, float *y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = la::bf16_to_f32(x[i]);
}

// mel [B][n_mels][frames] f32 -> out[b][1+t][c] (c_pad wide), zero borders / pad channels.
// 32x32 tile transpose through LDS: reads coalesced along frames, writes along channels.
template <typename TOut>
__global__ __launch_bounds__(256) void mel_to_rows_kernel(const float *mel, int64_t mbs, int64_t mrs, int n_mels,
                                                          int frames, TOut *out, int c_pad) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32;  // over frames + 2 rows (row index r = t + 1)
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = t0 + tx;  // r: output row (0 .. frames+1)
        const int t = r - 1;
        float v = 0.f;
        if (c < n_mels && t >= 0 && t < frames) v = mel[(int64_t)b * mbs + (int64_t)c * mrs + t];
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int r = t0 + i, c = c0 + tx;
        if (r < frames + 2 && c < c_pad) {
            TOut *o = out + ((int64_t)b * (frames + 2) + r) * c_pad + c;
            la::Elem<TOut>::store(o, tile[tx][i]);
        }
    }
}

__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t *tokens, int n_tok, const float *emb, int n_vocab,
                                                           const float *pos, int d, float *x) {
    const int row = blockIdx.x;  // b * n_tok + i
    const int i = row % n_tok;
    int64_t tk = tokens[row];
    tk = tk < 0 ? 0 : (tk >= n_vocab ? n_vocab - 1 : tk);
    const float4 *e = reinterpret_cast<const float4 *>(emb + tk * d);
    const float4 *pp = reinterpret_cast<const float4 *>(pos + (int64_t)i * d);
    float4 *o = reinterpret_cast<float4 *>(x + (int64_t)row * d);
    for (int c = threadIdx.x; c < d / 4; c += 256) {
        const float4 a = e[c], b = pp[c];
        o[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

}  // namespace

extern "C" int la_embed_tokens(const int64_t *tokens, int32_t batch, int32_t n_tok, const float *token_embedding,
                               int32_t n_vocab, const float *positional_embedding, int32_t d, float *x, void *stream_) {
    if (batch == 0 || n_tok == 0) return LA_OK;
    LA_CHECK_ARG(tokens && token_embedding && positional_embedding && x && batch > 0 && n_tok > 0 && n_vocab > 0 && d > 0 && d % 4 == 0,
                 "embed_tokens: bad arguments");
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(batch * n_tok), dim3(256), 0, (hipStream_t)stream_, tokens, n_tok,
                       token_embedding, n_vocab, positional_embedding, d, x);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_emissions_from_logits(const float *logits, int64_t batch_stride, int64_t row_stride, int32_t batch,
                                        int32_t frames, int32_t vocab, int32_t variant, const int32_t *labels,
                                        int32_t labels_stride, const int32_t *n_labels, int32_t max_labels, float *em,
                                        int64_t em_batch_stride, int64_t em_row_stride, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0 || frames == 0) return LA_OK;
    LA_CHECK_ARG(logits && labels && n_labels && em, "emissions_from_logits: null pointer");
    LA_CHECK_ARG(batch > 0 && frames > 0 && max_labels > 0, "emissions_from_logits: bad sizes");
    LA_CHECK_ARG(variant == LA_VARIANT_CTC ? vocab >= 3 : vocab >= 2, "emissions_from_logits: vocab too small");
    LA_CHECK_ARG(em_row_stride >= max_labels + 1, "emissions_from_logits: em_row_stride < max_labels + 1");
    la::TimerScope ts("emissions_from_logits", stream);
    hipLaunchKernelGGL(emissions_from_logits_kernel, dim3(batch * frames), dim3(256), 0, stream, logits, batch_stride,
                       row_stride, frames, vocab, variant, labels, labels_stride, n_labels, max_labels, em,
                       em_batch_stride, em_row_stride);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_layernorm(const float *x, int64_t ldx, int32_t M, int32_t d, const float *gamma, const float *beta,
                            void *y, int64_t ldy, int32_t out_dtype, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M == 0) return LA_OK;
    LA_CHECK_ARG(x && gamma && beta && y, "layernorm: null pointer");
    LA_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 64 * 4 * 8 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: d=%d unsupported", d);
    LA_CHECK_ARG(out_dtype == LA_F32 || out_dtype == LA_BF16 || out_dtype == LA_F16, "layernorm: bad dtype");
    la::TimerScope ts("layernorm", stream);
    const dim3 grid(la::cdiv(M, 4)), block(256);
    if (out_dtype == LA_F32)
        hipLaunchKernelGGL((layernorm_kernel<float, 8>), grid, block, 0, stream, x, ldx, M, d, gamma, beta, (float *)y, ldy);
    else if (out_dtype == LA_F16)
        hipLaunchKernelGGL((layernorm_kernel<la::f16_t, 8>), grid, block, 0, stream, x, ldx, M, d, gamma, beta, (la::f16_t *)y, ldy);
    else
        hipLaunchKernelGGL((layernorm_kernel<bf16_t, 8>), grid, block, 0, stream, x, ldx, M, d, gamma, beta, (bf16_t *)y, ldy);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

template <typename T16>
static int layernorm_split_launch(const void *hi, const void *lo, int64_t ldx, int M, int d, const float *gamma, const float *beta, void *y,
                                  int64_t ldy, int out_dtype, hipStream_t stream) {
    const dim3 grid(la::cdiv(M, 4)), block(256);
    const unsigned short *h = reinterpret_cast<const unsigned short *>(hi);
    const unsigned char *l = reinterpret_cast<const unsigned char *>(lo);
    if (out_dtype == LA_F32)
        hipLaunchKernelGGL((layernorm_split_kernel<T16, float, 8>), grid, block, 0, stream, h, l, ldx, M, d, gamma, beta, (float *)y, ldy);
    else if (out_dtype == LA_F16)
        hipLaunchKernelGGL((layernorm_split_kernel<T16, la::f16_t, 8>), grid, block, 0, stream, h, l, ldx, M, d, gamma, beta, (la::f16_t *)y, ldy);
    else
        hipLaunchKernelGGL((layernorm_split_kernel<T16, bf16_t, 8>), grid, block, 0, stream, h, l, ldx, M, d, gamma, beta, (bf16_t *)y, ldy);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// LayerNorm of the split residual stream (hi `dtype` [M][ldx] + lo uint8 [M][ldx]); ln_post of the 16-bit encoder.
extern "C" int la_layernorm_split(int32_t dtype, const void *hi, const void *lo, int64_t ldx, int32_t M, int32_t d, const float *gamma,
                                  const float *beta, void *y, int64_t ldy, int32_t out_dtype, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M == 0) return LA_OK;
    LA_CHECK_ARG(hi && lo && gamma && beta && y, "layernorm_split: null pointer");
    LA_CHECK_ARG(dtype == LA_BF16 || dtype == LA_F16, "layernorm_split: the stream's hi part is bf16 or f16");
    LA_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 64 * 4 * 8 && ldx % 4 == 0 && ldy % 4 == 0 && (uintptr_t)hi % 8 == 0 && (uintptr_t)lo % 4 == 0,
                 "layernorm_split: d=%d unsupported or misaligned rows", d);
    LA_CHECK_ARG(out_dtype == LA_F32 || out_dtype == LA_BF16 || out_dtype == LA_F16, "layernorm_split: bad dtype");
    la::TimerScope ts("layernorm", stream);
    return dtype == LA_F16 ? layernorm_split_launch<la::f16_t>(hi, lo, ldx, M, d, gamma, beta, y, ldy, out_dtype, stream)
                           : layernorm_split_launch<bf16_t>(hi, lo, ldx, M, d, gamma, beta, y, ldy, out_dtype, stream);
}

extern "C" int la_row_stats16(int32_t dtype, const void *x, int64_t ldx, int32_t M, int32_t d, float eps, float *stats, void *stream_) {
    if (M == 0) return LA_OK;
    LA_CHECK_ARG(x && stats && M > 0 && d > 0, "row_stats16: bad arguments");
    LA_CHECK_ARG(dtype == LA_BF16 || dtype == LA_F16, "row_stats16: 16-bit rows only");
    LA_CHECK_ARG(d % 8 == 0 && d <= 64 * 8 * 4 && ldx % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)stats % 8 == 0),
                 "row_stats16: d must be a multiple of 8 and <= 2048, rows 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    la::TimerScope ts("layernorm", stream);
    const dim3 grid(la::cdiv(M, 4)), block(256);
    if (dtype == LA_F16)
        hipLaunchKernelGGL((row_stats16_kernel<la::f16_t, 4>), grid, block, 0, stream, (const la::f16_t *)x, ldx, M, d, eps, (float2 *)stats);
    else
        hipLaunchKernelGGL((row_stats16_kernel<bf16_t, 4>), grid, block, 0, stream, (const bf16_t *)x, ldx, M, d, eps, (float2 *)stats);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_cast_f32_to_bf16(const float *x, void *y, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && y && n > 0, "cast: bad arguments");
    LA_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0), "cast: pointers must be 16/8-byte aligned");
    const int grid = (int)std::min<int64_t>(2048, la::cdiv(n, 1024));
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, x, (bf16_t *)y, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_cast_bf16_to_f32(const void *x, float *y, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && y && n > 0, "cast: bad arguments");
    const int grid = (int)std::min<int64_t>(2048, la::cdiv(n, 256));
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)x, y, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_mel_to_rows(const float *mel, int64_t mel_batch_stride, int64_t mel_row_stride, int32_t batch,
                              int32_t n_mels, int32_t frames, void *out, int32_t c_pad, int32_t dtype, void *stream_) {
    if (batch == 0) return LA_OK;
    LA_CHECK_ARG(mel && out && batch > 0 && n_mels > 0 && frames > 0 && c_pad >= n_mels, "mel_to_rows: bad arguments");
    LA_CHECK_ARG(dtype == LA_F32 || dtype == LA_BF16 || dtype == LA_F16, "mel_to_rows: bad dtype");
    const dim3 grid(la::cdiv(frames + 2, 32), la::cdiv(c_pad, 32), batch), block(256);
    hipStream_t stream = (hipStream_t)stream_;
    if (dtype == LA_F32)
        hipLaunchKernelGGL((mel_to_rows_kernel<float>), grid, block, 0, stream, mel, mel_batch_stride, mel_row_stride,
                           n_mels, frames, (float *)out, c_pad);
    else if (dtype == LA_F16)
        hipLaunchKernelGGL((mel_to_rows_kernel<la::f16_t>), grid, block, 0, stream, mel, mel_batch_stride, mel_row_stride,
                           n_mels, frames, (la::f16_t *)out, c_pad);
    else
        hipLaunchKernelGGL((mel_to_rows_kernel<bf16_t>), grid, block, 0, stream, mel, mel_batch_stride, mel_row_stride,
                           n_mels, frames, (bf16_t *)out, c_pad);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// greedy token choice: out[r] = index of the first maximum of x[r][0..cols)   (torch.argmax semantics on ties: lowest index)
namespace {
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float *x, int64_t ld, int cols, int64_t *out) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const float *r = x + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = r[c];
        if (v > best || idx == 0x7fffffff) { best = v; idx = c; }   // c ascends per thread: strict > keeps the first maximum
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
    }
}
}  // namespace

extern "C" int la_argmax_rows_f32(const float *x, int64_t ld, int32_t rows, int32_t cols, int64_t *out, void *stream_) {
    if (rows == 0) return LA_OK;
    LA_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "argmax_rows: bad arguments");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream_, x, ld, cols, out);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

// top-k of each row plus the row's log-sum-exp (beam search: log-probabilities of the k best next tokens are
// value - lse).  k <= 8; one workgroup per row, k rounds of a masked arg-max (first maximum on ties).
namespace {
__global__ __launch_bounds__(256) void topk_rows_kernel(const float *x, int64_t ld, int cols, int k, float *vals, int64_t *idx,
                                                       float *lse) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int chosen[8];
    __shared__ float red[4];
    const float *r = x + (int64_t)blockIdx.x * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int round = 0; round < k; ++round) {
        float best = -INFINITY;
        int at = 0x7fffffff;
        for (int c = threadIdx.x; c < cols; c += 256) {
            bool taken = false;
            for (int j = 0; j < round; ++j) taken |= chosen[j] == c;
            const float v = r[c];
            if (!taken && (v > best || at == 0x7fffffff)) { best = v; at = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(at, o);
            if (oi != 0x7fffffff && (at == 0x7fffffff || ov > best || (ov == best && oi < at))) { best = ov; at = oi; }
        }
        if (lane == 0) { bv[wave] = best; bi[wave] = at; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (bi[w] != 0x7fffffff && (at == 0x7fffffff || bv[w] > best || (bv[w] == best && bi[w] < at))) { best = bv[w]; at = bi[w]; }
            chosen[round] = at;
            vals[(int64_t)blockIdx.x * k + round] = best;
            idx[(int64_t)blockIdx.x * k + round] = at == 0x7fffffff ? 0 : at;
        }
        __syncthreads();
    }
    // log-sum-exp around the row maximum (= the first chosen value)
    const float m = vals[(int64_t)blockIdx.x * k];
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += expf(r[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) lse[blockIdx.x] = m + logf((red[0] + red[1]) + (red[2] + red[3]));
}
}  // namespace

extern "C" int la_topk_rows_f32(const float *x, int64_t ld, int32_t rows, int32_t cols, int32_t k, float *vals, int64_t *idx,
                                float *lse, void *stream_) {
    if (rows == 0) return LA_OK;
    LA_CHECK_ARG(x && vals && idx && lse && rows > 0 && cols > 0 && ld >= cols && k >= 1 && k <= 8 && k <= cols, "topk_rows: bad arguments");
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream_, x, ld, cols, k, vals, idx, lse);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
