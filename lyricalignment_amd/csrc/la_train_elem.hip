// la_train_elem.hip -- HBM-bound helpers of the fine-tune backward pass (float32, like the reference's training):
// 2-D transpose with zero padding (turns dX = dY W and dW = dY^T X into the K-contiguous "NT" GEMMs la_gemm runs),
// column sums (bias gradients), Mish backward, masked scaling (inter-layer GRU dropout, module/align_model.py:23-28).
#include "la_common.h"

namespace {

// out[c][r] = in[r][c] for r < rows, c < cols; out is [cols_pad][ld_out] and everything outside is zero-filled
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float *in, int64_t ld_in, int rows, int cols, float *out,
                                                            int64_t ld_out, int out_rows, int out_cols, int64_t bs_in,
                                                            int64_t bs_out) {
    __shared__ float tile[32][33];
    in += (int64_t)blockIdx.z * bs_in;
    out += (int64_t)blockIdx.z * bs_out;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(int64_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;   // out row = c, out col = r
        if (c < out_rows && r < out_cols) out[(int64_t)c * ld_out + r] = tile[tx][i];
    }
}

// out[c] = sum_r in[r][c] in two deterministic passes: (64-column tile x row chunk) workgroups write double partials,
// then one thread per column adds the chunks in order.  Fills the chip for tall-skinny inputs (3000 x 1024: 752 groups).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *in, int64_t ld, int rows, int cols, int rows_per_chunk,
                                                             double *part) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    double s = 0.0;
    if (c < cols)
        for (int r = r0 + w; r < r1; r += 4) s += (double)in[(int64_t)r * ld + c];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols)
        part[(int64_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void colsum_final_kernel(const double *part, int chunks, int cols, float *out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * cols + c];
    out[c] = (float)s;
}

// dx = dy * mish'(x);  mish(x) = x tanh(sp), sp = softplus(x):  mish' = tanh(sp) + x (1 - tanh(sp)^2) sigmoid(x)
__global__ void mish_bwd_kernel(const float *x, const float *dy, float *dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float sp = v > 20.0f ? v : log1pf(expf(v));
        const float th = tanhf(sp);
        const float sg = 1.0f / (1.0f + expf(-v));
        dx[i] = dy[i] * (th + v * (1.0f - th * th) * sg);
    }
}

__global__ void mish_fwd_kernel(const float *x, float *y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = la::mish(x[i]);
}

// y = x * mask * scale (mask in {0,1} as uint8)
__global__ void mask_scale_kernel(const float *x, const unsigned char *mask, float scale, float *y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = mask[i] ? x[i] * scale : 0.f;
}

// exact-erf GELU forward / backward on float32 buffers (whisper's nn.GELU): gelu'(x) = Phi(x) + x phi(x)
__global__ void gelu_fwd_kernel(const float *x, float *y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = la::gelu_erf(x[i]);
}
__global__ void gelu_bwd_kernel(const float *x, const float *dy, float *dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        dx[i] = dy[i] * la::gelu_erf_grad(x[i]);
    }
}
__global__ void scale_kernel(const float *x, float alpha, float *y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = alpha * x[i];
}
// y = a + b
__global__ void add_kernel(const float *a, const float *b, float *y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// LayerNorm backward, one wave per row (row statistics recomputed):  xhat = (x - mean) rstd,  g = dy * gamma,
// dx = rstd (g - mean(g) - xhat mean(g xhat));  also writes dy * xhat (its column sum is dgamma; colsum(dy) is dbeta)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *x, const float *dy, const float *gamma, int M, int d,
                                                            float *dx, float *dy_xhat) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float *xr = x + (int64_t)row * d, *gr = dy + (int64_t)row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += xr[c];
    const float mean = wsum(s) / (float)d;
    float sq = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = xr[c] - mean; sq += t * t; }
    const float rstd = 1.0f / sqrtf(wsum(sq) / (float)d + 1e-5f);
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float xh = (xr[c] - mean) * rstd, g = gr[c] * gamma[c];
        sg += g; sgx += g * xh;
    }
    const float mg = wsum(sg) / (float)d, mgx = wsum(sgx) / (float)d;
    for (int c = lane; c < d; c += 64) {
        const float xh = (xr[c] - mean) * rstd, g = gr[c] * gamma[c];
        dx[(int64_t)row * d + c] = rstd * (g - mg - xh * mgx);
        dy_xhat[(int64_t)row * d + c] = gr[c] * xh;
    }
}

// The same with the two column sums taken on the way (dgamma = sum_r dy xhat, dbeta = sum_r dy): a row lives in registers (NV float4 per
// lane, d = 256 NV), x and dy are read once, dy * xhat is never written.  A workgroup takes 4 x rpw rows (rpw <= 32 per wave, fewer when the input is short: the host picks it so that a few hundred
// workgroups exist); a lane adds its columns'
// terms over its wave's rows in float32 (<= 32 terms), the four waves' sums are added in float64 in wave order and written per workgroup;
// ln_sums_final_kernel adds the workgroups in order (deterministic).  137 us + two column-sum passes -> one pass.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_sums_kernel(const float *x, const float *dy, const float *gamma, const float *resid, int M,
                                                                 int rpw, float *dx, double *part) {
    constexpr int d = 256 * NV;
    __shared__ float red[2][4][d];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * rpw;          // rpw rows per wave (host: 32 down to 2, so that a few hundred workgroups exist)
    float4 gm[NV], sg_[NV], sb_[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gm[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        sg_[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        sb_[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = 0; i < rpw; ++i) {
        const int row = row0 + i;
        if (row >= M) break;                                   // wave-uniform
        const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * d), *gr = reinterpret_cast<const float4 *>(dy + (int64_t)row * d);
        float4 xv[NV], gv[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { xv[j] = xr[lane + 64 * j]; gv[j] = gr[lane + 64 * j]; }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
        const float mean = wsum(s) / (float)d;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            xv[j].x -= mean; xv[j].y -= mean; xv[j].z -= mean; xv[j].w -= mean;
            sq += (xv[j].x * xv[j].x + xv[j].y * xv[j].y) + (xv[j].z * xv[j].z + xv[j].w * xv[j].w);
        }
        const float rstd = 1.0f / sqrtf(wsum(sq) / (float)d + 1e-5f);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            xv[j].x *= rstd; xv[j].y *= rstd; xv[j].z *= rstd; xv[j].w *= rstd;          // xhat
            const float4 g = make_float4(gv[j].x * gm[j].x, gv[j].y * gm[j].y, gv[j].z * gm[j].z, gv[j].w * gm[j].w);
            sg += (g.x + g.y) + (g.z + g.w);
            sgx += (g.x * xv[j].x + g.y * xv[j].y) + (g.z * xv[j].z + g.w * xv[j].w);
        }
        const float mg = wsum(sg) / (float)d, mgx = wsum(sgx) / (float)d;
        float4 *dr = reinterpret_cast<float4 *>(dx + (int64_t)row * d);
        const float4 *rr = resid ? reinterpret_cast<const float4 *>(resid + (int64_t)row * d) : nullptr;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 xh = xv[j], gy = gv[j];
            float4 o = make_float4(rstd * (gy.x * gm[j].x - mg - xh.x * mgx), rstd * (gy.y * gm[j].y - mg - xh.y * mgx),
                                   rstd * (gy.z * gm[j].z - mg - xh.z * mgx), rstd * (gy.w * gm[j].w - mg - xh.w * mgx));
            if (rr) {                                            // the gradient that reaches x past the block (residual connection)
                const float4 t = rr[lane + 64 * j];
                o = make_float4(t.x + o.x, t.y + o.y, t.z + o.z, t.w + o.w);
            }
            dr[lane + 64 * j] = o;
            sg_[j].x += gy.x * xh.x; sg_[j].y += gy.y * xh.y; sg_[j].z += gy.z * xh.z; sg_[j].w += gy.w * xh.w;
            sb_[j].x += gy.x; sb_[j].y += gy.y; sb_[j].z += gy.z; sb_[j].w += gy.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        reinterpret_cast<float4 *>(red[0][wave])[lane + 64 * j] = sg_[j];
        reinterpret_cast<float4 *>(red[1][wave])[lane + 64 * j] = sb_[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += 256) {
        const int which = c / d, col = c % d;
        part[((int64_t)blockIdx.x * 2 + which) * d + col] =
            ((double)red[which][0][col] + (double)red[which][1][col]) + ((double)red[which][2][col] + (double)red[which][3][col]);
    }
}
__global__ __launch_bounds__(256) void ln_sums_final_kernel(const double *part, int blocks, int d, float *dgamma, float *dbeta) {
    __shared__ double red[4][64];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;                      // over 2 d columns: [dgamma | dbeta]
    double s = 0.0;
    if (c < 2 * d)
        for (int k = ph; k < blocks; k += 4) s += part[(int64_t)k * 2 * d + c];
    red[ph][cx] = s;
    __syncthreads();
    if (ph == 0 && c < 2 * d) {
        const float v = (float)((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]));
        if (c < d) dgamma[c] = v; else dbeta[c - d] = v;
    }
}

// in-place row softmax over the first `cols` columns of rows with pitch ld (pad columns are set to 0)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *s, int64_t ld, int64_t rows, int cols_all, int causal_q_len) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float *r = s + row * ld;
    // causal: query i (= row mod q_len) sees keys 0 .. i + (kv_len - q_len); masked entries become exact zeros
    int cols = cols_all;
    if (causal_q_len > 0) cols = min(cols_all, max(1, (int)(row % causal_q_len) + 1 + (cols_all - causal_q_len)));
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, r[c]);
    m = wmax(m);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) { const float e = expf(r[c] - m); r[c] = e; sum += e; }
    const float inv = 1.0f / wsum(sum);
    for (int c = lane; c < cols; c += 64) r[c] *= inv;
    for (int c = cols + lane; c < ld; c += 64) r[c] = 0.f;
}
// dS = P * (dP - sum_c dP P), written over dP
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float *p, float *dp, int64_t ld, int64_t rows, int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *pr = p + row * ld;
    float *dr = dp + row * ld;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot += pr[c] * dr[c];
    dot = wsum(dot);
    for (int c = lane; c < cols; c += 64) dr[c] = pr[c] * (dr[c] - dot);
    for (int c = cols + lane; c < ld; c += 64) dr[c] = 0.f;
}

// overlap-add of the k=3 convolution's column gradients: out[b][r][c] = sum over (t, tap) with t*stride + tap == r of
// dcols[b][t][tap][c];  out is [B][rows_out][C] (the zero-bordered channels-last input buffer of the conv-as-GEMM view)
__global__ void col2im3_kernel(const float *dcols, int T_out, int stride, int C, float *out, int rows_out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int r = (int)((i / C) % rows_out);
        const int64_t b = i / ((int64_t)C * rows_out);
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int num = r - tap;
            if (num >= 0 && num % stride == 0) {
                const int t = num / stride;
                if (t < T_out) acc += dcols[((b * T_out + t) * 3 + tap) * (int64_t)C + c];
            }
        }
        out[i] = acc;
    }
}

// gradient of x[b][i][:] = tok_emb[tokens[b][i]][:] + pos[i][:]:  dtok rows accumulate (atomics: a token may repeat),
// dpos[i] = sum over the batch (written, deterministic order)
// token ids are clamped to [0, n_vocab) exactly as embed_tokens_kernel clamps them in the forward pass (an out-of-range id in user
// data must not become an out-of-bounds atomic)
__global__ void embed_bwd_kernel(const float *dx, const int64_t *tokens, int B, int n, int d, int n_vocab, float *dtok, float *dpos) {
    const int64_t total = (int64_t)n * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d), pos_i = (int)(i / d);
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float g = dx[((int64_t)b * n + pos_i) * d + c];
            acc += g;
            int64_t tk = tokens[(int64_t)b * n + pos_i];
            tk = tk < 0 ? 0 : (tk >= n_vocab ? n_vocab - 1 : tk);
            atomicAdd(dtok + tk * d + c, g);
        }
        dpos[i] = acc;
    }
}

// F.cross_entropy(logits [R][V], target [R], ignore_index = -100, reduction = 'mean') -- three small passes
__global__ __launch_bounds__(256) void ce_row_kernel(const float *logits, int64_t ld, int V, const int64_t *target, float *row_lse,
                                                      float *row_loss) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float *l = logits + (int64_t)r * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, l[c]);
    m = wmax(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += expf(l[c] - m);
    s = wsum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lse = m + logf(red[0] + red[1] + red[2] + red[3]);
        row_lse[r] = lse;
        const int64_t t = target[r];
        row_loss[r] = (t >= 0 && t < V) ? lse - l[t] : 0.f;
    }
}
__global__ __launch_bounds__(64) void ce_reduce_kernel(const float *row_loss, const int64_t *target, int R, int V, float *out2) {
    double s = 0.0;
    int cnt = 0;
    for (int r = threadIdx.x; r < R; r += 64) {
        const int64_t t = target[r];
        if (t >= 0 && t < V) { s += (double)row_loss[r]; ++cnt; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); cnt += __shfl_xor(cnt, o); }
    if (threadIdx.x == 0) {
        out2[0] = cnt > 0 ? (float)(s / cnt) : NAN;      // torch returns nan for an all-ignored batch
        out2[1] = cnt > 0 ? 1.0f / (float)cnt : 0.f;
    }
}
__global__ void ce_grad_kernel(const float *logits, int64_t ld, int V, const int64_t *target, const float *row_lse, const float *out2,
                               float scale, float *dlogits, int64_t ld_d, int64_t n) {
    const float k = out2[1] * scale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % V);
        const int64_t r = i / V;
        const int64_t t = target[r];
        float g = 0.f;
        if (t >= 0 && t < V) g = (expf(logits[r * ld + c] - row_lse[r]) - (c == t ? 1.f : 0.f)) * k;
        dlogits[r * ld_d + c] = g;
    }
}

inline int ew_grid(int64_t n) { return (int)std::min<int64_t>(4096, la::cdiv(n, 256)); }

}  // namespace

extern "C" int la_transpose_pad_f32(const float *in, int64_t ld_in, int32_t rows, int32_t cols, float *out, int64_t ld_out,
                                    int32_t out_rows, int32_t out_cols, void *stream_) {
    LA_CHECK_ARG(in && out && rows > 0 && cols > 0 && out_rows >= cols && out_cols >= rows && ld_out >= out_cols && ld_in >= cols,
                 "transpose_pad: bad arguments");
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(la::cdiv(out_cols, 32), la::cdiv(out_rows, 32), 1), dim3(256), 0,
                       (hipStream_t)stream_, in, ld_in, rows, cols, out, ld_out, out_rows, out_cols, (int64_t)0, (int64_t)0);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_transpose_pad_batched_f32(const float *in, int64_t ld_in, int64_t batch_stride_in, int32_t rows, int32_t cols,
                                            float *out, int64_t ld_out, int64_t batch_stride_out, int32_t out_rows,
                                            int32_t out_cols, int32_t batch, void *stream_) {
    LA_CHECK_ARG(in && out && rows > 0 && cols > 0 && batch > 0 && out_rows >= cols && out_cols >= rows && ld_out >= out_cols && ld_in >= cols,
                 "transpose_pad_batched: bad arguments");
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(la::cdiv(out_cols, 32), la::cdiv(out_rows, 32), batch), dim3(256), 0,
                       (hipStream_t)stream_, in, ld_in, rows, cols, out, ld_out, out_rows, out_cols, batch_stride_in, batch_stride_out);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_colsum_f32(const float *in, int64_t ld, int32_t rows, int32_t cols, float *out, void *stream_) {
    LA_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    const int chunks = std::max(1, std::min(64, la::cdiv(rows, 64)));
    const int rows_per_chunk = la::cdiv(rows, chunks);
    double *part = static_cast<double *>(la::stream_scratch(st, la::SCRATCH_COLSUM, sizeof(double) * (size_t)chunks * cols));   // per stream
    if (!part) { la::set_error("colsum: scratch allocation failed"); return LA_EHIP; }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(la::cdiv(cols, 64), chunks), dim3(256), 0, st, in, ld, rows, cols, rows_per_chunk, part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(la::cdiv(cols, 256)), dim3(256), 0, st, part, chunks, cols, out);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_mish_f32(const float *x, float *y, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && y && n > 0, "mish: bad arguments");
    hipLaunchKernelGGL(mish_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, x, y, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_mish_bwd_f32(const float *x, const float *dy, float *dx, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && dy && dx && n > 0, "mish_bwd: bad arguments");
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, x, dy, dx, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_mask_scale_f32(const float *x, const unsigned char *mask, float scale, float *y, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && mask && y && n > 0, "mask_scale: bad arguments");
    hipLaunchKernelGGL(mask_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, x, mask, scale, y, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_gelu_f32(const float *x, float *y, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && y && n > 0, "gelu: bad arguments");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, x, y, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
extern "C" int la_gelu_bwd_f32(const float *x, const float *dy, float *dx, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && dy && dx && n > 0, "gelu_bwd: bad arguments");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, x, dy, dx, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
extern "C" int la_add_f32(const float *a, const float *b, float *y, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(a && b && y && n > 0, "add: bad arguments");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, a, b, y, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
extern "C" int la_layernorm_bwd_f32(const float *x, const float *dy, const float *gamma, int32_t M, int32_t d, float *dx,
                                    float *dy_xhat, void *stream_) {
    LA_CHECK_ARG(x && dy && gamma && dx && dy_xhat && M > 0 && d > 0, "layernorm_bwd: bad arguments");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(la::cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream_, x, dy, gamma, M, d, dx, dy_xhat);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
extern "C" int la_colsum_f32(const float *in, int64_t ld, int32_t rows, int32_t cols, float *out, void *stream_);

// LayerNorm backward with its parameter gradients: dx (+ residual, if given: the gradient that bypasses the block), dgamma [d] = sum_r dy xhat,
// dbeta [d] = sum_r dy.  d = 256, 512, ... 2048 with
// 16-byte aligned rows: one pass (layernorm_bwd_sums_kernel); otherwise la_layernorm_bwd_f32 into `scratch` (M x d floats, required then)
// and two la_colsum_f32.
extern "C" int la_add_f32(const float *a, const float *b, float *y, int64_t n, void *stream_);

extern "C" int la_layernorm_bwd_sums_f32(const float *x, const float *dy, const float *gamma, const float *residual, int32_t M, int32_t d, float *dx,
                                         float *dgamma, float *dbeta, float *scratch, void *stream_) {
    LA_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta && M > 0 && d > 0, "layernorm_bwd_sums: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    const bool fast = d % 256 == 0 && d <= 2048 && ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)residual) % 16 == 0;
    if (!fast) {
        LA_CHECK_ARG(scratch, "layernorm_bwd_sums: this width needs the M x d scratch buffer");
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(la::cdiv(M, 4)), dim3(256), 0, st, x, dy, gamma, M, d, dx, scratch);
        LA_LAUNCH_CHECK();
        int rc = la_colsum_f32(scratch, d, M, d, dgamma, stream_);
        if (rc == LA_OK) rc = la_colsum_f32(dy, d, M, d, dbeta, stream_);
        if (rc == LA_OK && residual) rc = la_add_f32(residual, dx, dx, (int64_t)M * d, stream_);
        return rc;
    }
    // rows per wave: 32 for tall inputs, fewer when that would leave the chip short of workgroups (3000 rows, a micro-batch of the
    // reference's own loop: 24 workgroups of 128 rows took 93 us, the time of 24000 rows)
    int rpw = 32;
    while (rpw > 2 && la::cdiv(M, 4 * rpw) < 512) rpw >>= 1;
    const int blocks = la::cdiv(M, 4 * rpw);
    double *part = static_cast<double *>(la::stream_scratch(st, la::SCRATCH_COLSUM, sizeof(double) * (size_t)blocks * 2 * d));
    if (!part) { la::set_error("layernorm_bwd_sums: scratch allocation failed"); return LA_EHIP; }
    switch (d / 256) {
#define LA_LN_CASE(nv) case nv: hipLaunchKernelGGL(layernorm_bwd_sums_kernel<nv>, dim3(blocks), dim3(256), 0, st, x, dy, gamma, residual, M, rpw, dx, part); break;
        LA_LN_CASE(1) LA_LN_CASE(2) LA_LN_CASE(3) LA_LN_CASE(4) LA_LN_CASE(5) LA_LN_CASE(6) LA_LN_CASE(7) LA_LN_CASE(8)
#undef LA_LN_CASE
    }
    LA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_sums_final_kernel, dim3(la::cdiv(2 * d, 64)), dim3(256), 0, st, part, blocks, d, dgamma, dbeta);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
extern "C" int la_softmax_rows_f32(float *s, int64_t ld, int64_t rows, int32_t cols, int32_t causal_q_len, void *stream_) {
    LA_CHECK_ARG(s && rows > 0 && cols > 0 && ld >= cols && causal_q_len >= 0, "softmax_rows: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(la::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream_, s, ld, rows, cols, causal_q_len);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
extern "C" int la_softmax_bwd_rows_f32(const float *p, float *dp, int64_t ld, int64_t rows, int32_t cols, void *stream_) {
    LA_CHECK_ARG(p && dp && rows > 0 && cols > 0 && ld >= cols, "softmax_bwd_rows: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(la::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream_, p, dp, ld, rows, cols);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
extern "C" int la_col2im3_f32(const float *dcols, int32_t batch, int32_t t_out, int32_t stride, int32_t channels, float *out,
                              int32_t rows_out, void *stream_) {
    LA_CHECK_ARG(dcols && out && batch > 0 && t_out > 0 && stride > 0 && channels > 0 && rows_out >= (t_out - 1) * stride + 3,
                 "col2im3: bad arguments");
    const int64_t n = (int64_t)batch * rows_out * channels;
    hipLaunchKernelGGL(col2im3_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, dcols, t_out, stride, channels, out, rows_out, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_scale_f32(const float *x, float alpha, float *y, int64_t n, void *stream_) {
    if (n == 0) return LA_OK;
    LA_CHECK_ARG(x && y && n > 0, "scale: bad arguments");
    hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream_, x, alpha, y, n);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_embed_tokens_bwd_f32(const float *dx, const int64_t *tokens, int32_t batch, int32_t n, int32_t d, int32_t n_vocab,
                                       float *dtok, float *dpos, void *stream_) {
    LA_CHECK_ARG(dx && tokens && dtok && dpos && batch > 0 && n > 0 && d > 0 && n_vocab > 0, "embed_tokens_bwd: bad arguments");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(ew_grid((int64_t)n * d)), dim3(256), 0, (hipStream_t)stream_, dx, tokens, batch, n, d, n_vocab, dtok, dpos);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_cross_entropy_f32(const float *logits, int64_t ld, int32_t rows, int32_t vocab, const int64_t *target, float scale,
                                    float *loss2, float *row_ws, float *dlogits, int64_t ld_d, void *stream_) {
    LA_CHECK_ARG(logits && target && loss2 && row_ws && rows > 0 && vocab > 0 && ld >= vocab && (!dlogits || ld_d >= vocab),
                 "cross_entropy: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    float *row_lse = row_ws, *row_loss = row_ws + rows;
    hipLaunchKernelGGL(ce_row_kernel, dim3(rows), dim3(256), 0, st, logits, ld, vocab, target, row_lse, row_loss);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(64), 0, st, row_loss, target, rows, vocab, loss2);
    if (dlogits) {
        const int64_t n = (int64_t)rows * vocab;
        hipLaunchKernelGGL(ce_grad_kernel, dim3(ew_grid(n)), dim3(256), 0, st, logits, ld, vocab, target, row_lse, loss2, scale, dlogits, ld_d, n);
    }
    LA_LAUNCH_CHECK();
    return LA_OK;
}
