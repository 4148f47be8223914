// la_train_elem.hip -- HBM-bound helpers of the fine-tune backward pass (float32, like the reference's training):
// 2-D transpose with zero padding (turns dX = dY W and dW = dY^T X into the K-contiguous "NT" GEMMs la_gemm runs),
// column sums (bias gradients), Mish backward, masked scaling (inter-layer GRU dropout, module/align_model.py:23-28).
#include "la_common.h"

namespace {

// out[c][r] = in[r][c] for r < rows, c < cols; out is [cols_pad][ld_out] and everything outside is zero-filled
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float *in, int64_t ld_in, int rows, int cols, float *out,
                                                            int64_t ld_out, int out_rows, int out_cols) {
    __shared__ float tile[32][33];
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

// out[c] = sum_r in[r][c]; one workgroup per 64 columns, rows strided over 4 waves, double accumulation
__global__ __launch_bounds__(256) void colsum_kernel(const float *in, int64_t ld, int rows, int cols, float *out) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    double s = 0.0;
    if (c < cols)
        for (int r = w; r < rows; r += 4) s += (double)in[(int64_t)r * ld + c];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols) out[c] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
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

inline int ew_grid(int64_t n) { return (int)std::min<int64_t>(4096, la::cdiv(n, 256)); }

}  // namespace

extern "C" int la_transpose_pad_f32(const float *in, int64_t ld_in, int32_t rows, int32_t cols, float *out, int64_t ld_out,
                                    int32_t out_rows, int32_t out_cols, void *stream_) {
    LA_CHECK_ARG(in && out && rows > 0 && cols > 0 && out_rows >= cols && out_cols >= rows && ld_out >= out_cols && ld_in >= cols,
                 "transpose_pad: bad arguments");
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(la::cdiv(out_cols, 32), la::cdiv(out_rows, 32)), dim3(256), 0,
                       (hipStream_t)stream_, in, ld_in, rows, cols, out, ld_out, out_rows, out_cols);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_colsum_f32(const float *in, int64_t ld, int32_t rows, int32_t cols, float *out, void *stream_) {
    LA_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3(la::cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream_, in, ld, rows, cols, out);
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
