// la_f32x2.hip -- float32 matrix products on the f16 matrix pipe at float32 accuracy ("f16x2"): the Linear layers of the fine-tune
// step (train_multitask.py:325-326: loss.backward() through nn.Linear / F.linear in float32 -- forward and both backward products).
//
// gfx950 multiplies float32 operands at 1/16 of its 16-bit MFMA rate (v_mfma_f32_16x16x4_f32: 157 TFLOP/s against 2.5 PFLOP/s).  A
// float32 value scaled by a power of two s (per ROW of its matrix, so that the row's largest magnitude lands in [2^13, 2^14)) splits
// EXACTLY into two IEEE halves  x s = hi + lo + r,  hi = f16(x s),  lo = f16(x s - hi),  |r| <= 2^-22 |x s|  (elements more than
// 2^-17 below their row's maximum lose relative -- never absolute -- precision to half's subnormal range: |r| <= 2^-25), and
//     sum_k a_k w_k  =  (sa sw)^-1  sum_k (a_lo w_hi + a_hi w_lo + a_hi w_hi)  +  O(2^-22)
// -- three f16 products accumulated in float32 by v_mfma_f32_16x16x32_f16, 3/16 of the float32 pipe's cost.  Measured on the
// fine-tune shapes (profiles/r5_kbench_f32emu.txt, prototype by K-concatenation on the shipped kernel): 2.5-3.2 x faster than
// gemm_kernel<float>, error against a float64 product 2-3 x SMALLER than the float32 kernel's own (whose 16x16x4 chains round
// eight times as often per k as the 16x16x32 instruction).
//
//   la_split_f16x2     x [rows][cols] f32  ->  planes [rows][2][kp] f16 (hi plane, lo plane; kp >= cols, zero tail) + inverse scales [rows]
//   la_split_f16x2_t   the same of x^T: planes [cols][2][mp] + inverse scales [cols]  (weight-gradient operands: contraction over rows)
//   la_gemm_f16x2      C = epi((A W^T) o sa sw^T): the 256 x 256 f16 kernel over segmented K (la_gemm_pp.h mainloop_duo_seg_asm),
//                      scales applied in the epilogue; `slots` > 1 cuts K over batch slots (few tiles, long K), summed in a fixed order
#include <algorithm>

#include "la_gemm_core.h"
#include "la_gemm_params.h"

using namespace la::gemm;

namespace {

// power of two s with max |x| s in [2^13, 2^14); max == 0 (or not finite) -> 1.  Returns s, *inv = 1 / s (both exact).
__device__ __forceinline__ float x2_scale(float mx, float *inv) {
    if (!(mx > 0.f) || !(mx < 3.0e38f)) { *inv = 1.f; return 1.f; }
    int e;
    (void)frexpf(mx, &e);                    // mx = m 2^e, m in [0.5, 1)
    int sh = 14 - e;
    sh = sh > 126 ? 126 : (sh < -126 ? -126 : sh);
    *inv = ldexpf(1.f, -sh);
    return ldexpf(1.f, sh);
}

__device__ __forceinline__ void x2_split(float xs, unsigned short &hi, unsigned short &lo) {
    const _Float16 h = (_Float16)xs;
    const _Float16 l = (_Float16)(xs - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

// optional activation applied to the float32 values before they are split (act: 0 none, 1 exact-erf GELU): the MLP's hidden operand
// gelu(u) is never materialised in float32 -- the forward and the weight-gradient products split it straight from u
__device__ __forceinline__ float x2_act(float v, int act) { return act == 1 ? la::gelu_erf(v) : v; }

// one wave per row, four rows per workgroup: pass 1 = the row's largest magnitude, pass 2 = split + store.  Rows of up to 4096 columns
// (every operand of the encoder) stay in registers between the passes -- 16 float4 per lane, all loads in flight at once: one read of x
// (91 -> ~55 us per 24000 x 1024 operand: the two-pass form was bound by its two dependent round trips per wave, not by bytes); longer
// rows are read twice (the second time from L2).
// tmax (or NULL): 64 words that collect the largest magnitude of the whole operand (bit patterns, atomicMax; word = workgroup mod 64) --
// what a later TRANSPOSED split of the same operand scales with instead of making its own pass for column maxima
__global__ __launch_bounds__(256) void split_rows_kernel(const float *x, int64_t ldx, int rows, int cols, unsigned short *planes, int64_t kp,
                                                         float *inv_scale, int act, unsigned *tmax) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (int64_t)row * ldx;
    const bool vec = (ldx % 4 == 0) && ((uintptr_t)x % 16 == 0);
    const int c4 = vec ? cols / 4 : 0;
    unsigned short *hi = planes + (int64_t)row * 2 * kp, *lo = hi + kp;
    float mx = 0.f;
    constexpr int NV = 16;
    const bool cached = c4 <= 64 * NV;
    float4 v[NV];
    if (cached) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < c4) v[j] = reinterpret_cast<const float4 *>(xr)[i];
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (act) v[j] = make_float4(x2_act(v[j].x, act), x2_act(v[j].y, act), x2_act(v[j].z, act), x2_act(v[j].w, act));
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
        }
    } else {
        for (int i = lane; i < c4; i += 64) {
            const float4 w = reinterpret_cast<const float4 *>(xr)[i];
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(x2_act(w.x, act)), fabsf(x2_act(w.y, act))), fmaxf(fabsf(x2_act(w.z, act)), fabsf(x2_act(w.w, act)))));
        }
    }
    for (int i = c4 * 4 + lane; i < cols; i += 64) mx = fmaxf(mx, fabsf(x2_act(xr[i], act)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float inv;
    const float s = x2_scale(mx, &inv);
    // one atomic per row into 64 words that lie 128 B apart (all 64 in one or two cache lines made the rows' 24000 atomics queue on one
    // L2 channel: +34 us per operand, more than the pass they were to save)
    if (tmax && lane == 0) atomicMax(tmax + (row & 63) * 32, __float_as_uint(mx != mx ? __uint_as_float(0x7f800000u) : mx));
    if (cached) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < c4) {
                ushort4 h, l;
                x2_split(v[j].x * s, h.x, l.x); x2_split(v[j].y * s, h.y, l.y); x2_split(v[j].z * s, h.z, l.z); x2_split(v[j].w * s, h.w, l.w);
                reinterpret_cast<ushort4 *>(hi)[i] = h;
                reinterpret_cast<ushort4 *>(lo)[i] = l;
            }
        }
    } else {
        for (int i = lane; i < c4; i += 64) {
            const float4 w = reinterpret_cast<const float4 *>(xr)[i];
            ushort4 h, l;
            x2_split(x2_act(w.x, act) * s, h.x, l.x); x2_split(x2_act(w.y, act) * s, h.y, l.y); x2_split(x2_act(w.z, act) * s, h.z, l.z);
            x2_split(x2_act(w.w, act) * s, h.w, l.w);
            reinterpret_cast<ushort4 *>(hi)[i] = h;
            reinterpret_cast<ushort4 *>(lo)[i] = l;
        }
    }
    for (int i = c4 * 4 + lane; i < kp; i += 64) {
        unsigned short h = 0, l = 0;
        if (i < cols) x2_split(x2_act(xr[i], act) * s, h, l);
        hi[i] = h; lo[i] = l;
    }
    if (lane == 0) inv_scale[row] = inv;
}

// LayerNorm and the operand split in ONE pass over the rows (float32 inference on the f16x2 products, la_model.cpp): y = LN(x) gamma + beta is
// what the next Linear multiplies, so its planes are made while the row is in registers -- the float32 copy of y (4 + 4 bytes per element
// written and read back by la_layernorm + la_split_f16x2) never exists.  One wave per row, rows of up to 4096 columns; statistics as
// la_layernorm (two passes over the registers: mean, then the mean of squared deviations; eps 1e-5).
__global__ __launch_bounds__(256) void ln_split_rows_kernel(const float *x, int64_t ldx, int rows, int d, const float *gamma, const float *beta,
                                                            unsigned short *planes, int64_t kp, float *inv_scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + (int64_t)row * ldx;
    constexpr int NV = 16;
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < d) {
            v[j] = *reinterpret_cast<const float4 *>(xr + c);
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < d) {
            const float a = v[j].x - mean, b = v[j].y - mean, c2 = v[j].z - mean, d2 = v[j].w - mean;
            sq += (a * a + b * b) + (c2 * c2 + d2 * d2);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)d + 1e-5f);
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < d) {
            const float4 g = *reinterpret_cast<const float4 *>(gamma + c), bt = *reinterpret_cast<const float4 *>(beta + c);
            v[j] = make_float4((v[j].x - mean) * rstd * g.x + bt.x, (v[j].y - mean) * rstd * g.y + bt.y, (v[j].z - mean) * rstd * g.z + bt.z,
                               (v[j].w - mean) * rstd * g.w + bt.w);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float inv;
    const float s = x2_scale(mx, &inv);
    unsigned short *hi = planes + (int64_t)row * 2 * kp, *lo = hi + kp;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i * 4 < d) {
            ushort4 h, l;
            x2_split(v[j].x * s, h.x, l.x); x2_split(v[j].y * s, h.y, l.y); x2_split(v[j].z * s, h.z, l.z); x2_split(v[j].w * s, h.w, l.w);
            reinterpret_cast<ushort4 *>(hi)[i] = h;
            reinterpret_cast<ushort4 *>(lo)[i] = l;
        }
    }
    for (int i = d + lane; i < kp; i += 64) { hi[i] = 0; lo[i] = 0; }
    if (lane == 0) inv_scale[row] = inv;
}

// column maxima of |x| as ordered unsigned bit patterns (|x| >= 0: the float order is the integer order)
__global__ __launch_bounds__(256) void colmax_kernel(const float *x, int64_t ldx, int rows, int cols, int rows_per_block, unsigned *colmax, int act) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float mx = 0.f;
    if (col < cols)
        for (int r = r0 + ry; r < r1; r += 4) mx = fmaxf(mx, fabsf(x2_act(x[(int64_t)r * ldx + col], act)));
    red[ry][cx] = mx;
    __syncthreads();
    if (ry == 0 && col < cols) {
        mx = fmaxf(fmaxf(red[0][cx], red[1][cx]), fmaxf(red[2][cx], red[3][cx]));
        if (mx != mx) mx = __uint_as_float(0x7f800000u);            // a NaN column: scale 1 (x2_scale), the NaNs pass through the split
        atomicMax(colmax + col, __float_as_uint(mx));
    }
}

// the same with 16-byte loads: a lane owns four consecutive columns, a workgroup 256 columns x rows_per_block rows (four row phases of 64
// lanes: 1 KiB contiguous per row and phase, four rows in flight per lane)
// SUMS: also the column sums of x (the bias gradient that goes with a weight gradient dy^T x: the same pass over dy) as float64 partials
// per row block, part[blockIdx.y][col]; colsum_blocks_kernel adds the blocks in order (deterministic, as la_colsum_f32)
template <bool SUMS>
__global__ __launch_bounds__(256) void colmax4_kernel(const float *x, int64_t ldx, int rows, int cols, int rows_per_block, unsigned *colmax, int act,
                                                      double *part) {
    __shared__ float4 red[4][64];
    __shared__ double reds[SUMS ? 4 : 1][SUMS ? 64 : 1][4];
    double sm[4] = {0.0, 0.0, 0.0, 0.0};
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + cx * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
    auto take = [&](const float4 &v) {
        mx.x = fmaxf(mx.x, fabsf(x2_act(v.x, act))); mx.y = fmaxf(mx.y, fabsf(x2_act(v.y, act)));
        mx.z = fmaxf(mx.z, fabsf(x2_act(v.z, act))); mx.w = fmaxf(mx.w, fabsf(x2_act(v.w, act)));
        if constexpr (SUMS) { sm[0] += (double)v.x; sm[1] += (double)v.y; sm[2] += (double)v.z; sm[3] += (double)v.w; }
    };
    if (col < cols) {                                   // cols is a multiple of 4 here: the quad is whole
        const float *p = x + col;
        int r = r0 + ry;
        for (; r + 12 < r1; r += 16) {
            const float4 a = *reinterpret_cast<const float4 *>(p + (int64_t)r * ldx), b = *reinterpret_cast<const float4 *>(p + (int64_t)(r + 4) * ldx);
            const float4 c = *reinterpret_cast<const float4 *>(p + (int64_t)(r + 8) * ldx), d = *reinterpret_cast<const float4 *>(p + (int64_t)(r + 12) * ldx);
            take(a); take(b); take(c); take(d);
        }
        for (; r < r1; r += 4) take(*reinterpret_cast<const float4 *>(p + (int64_t)r * ldx));
    }
    red[ry][cx] = mx;
    if constexpr (SUMS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) reds[ry][cx][e] = sm[e];
    }
    __syncthreads();
    if (ry == 0 && col < cols) {
        if constexpr (SUMS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) part[(int64_t)blockIdx.y * cols + col + e] = (reds[0][cx][e] + reds[1][cx][e]) + (reds[2][cx][e] + reds[3][cx][e]);
        }
        const float4 a = red[0][cx], b = red[1][cx], c = red[2][cx], d = red[3][cx];
        float m[4] = {fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)), fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)),
                      fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w))};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (m[e] != m[e]) m[e] = __uint_as_float(0x7f800000u);
            atomicMax(colmax + col + e, __float_as_uint(m[e]));
        }
    }
}

// out[c] = sum of the row blocks' partials in a fixed order: 64 columns per workgroup, four threads per column take every fourth block each,
// their sums are added as ((0 + 1) + (2 + 3))
__global__ __launch_bounds__(256) void colsum_blocks_kernel(const double *part, int blocks, int cols, float *out) {
    __shared__ double red[8][32];
    const int cx = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s = 0.0;
    if (c < cols)
        for (int k = ph; k < blocks; k += 8) s += part[(int64_t)k * cols + c];
    red[ph][cx] = s;
    __syncthreads();
    if (ph == 0 && c < cols)
        out[c] = (float)(((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) + ((red[4][cx] + red[5][cx]) + (red[6][cx] + red[7][cx])));
}

// 64 (rows of x) x 64 (columns of x) tiles through LDS: read along the columns, written along the rows of x (= along the padded
// contraction dimension of the transposed planes)
// TMAX: one scale for the whole operand, from the 64 words a plain split of the same operand left (split_rows_kernel tmax), instead of one per
// column from a pass of its own.  SUMS: the column sums of x as float64 partials per 64-row block, part[blockIdx.y][col] (act = 0).
template <bool TMAX, bool SUMS>
__global__ __launch_bounds__(256) void split_transposed_kernel(const float *x, int64_t ldx, int rows, int cols, const unsigned *colmax,
                                                               unsigned short *planes, int64_t mp, float *inv_scale, int act, double *part) {
    __shared__ float tile[64][65];
    const int m0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + ty * 16 + i, k = k0 + tx;
            tile[ty * 16 + i][tx] = (m < rows && k < cols) ? x2_act(x[(int64_t)m * ldx + k], act) : 0.f;
        }
    }
    unsigned tm = 0;
    if constexpr (TMAX) {
        tm = colmax[(threadIdx.x & 63) * 32];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tm = max(tm, (unsigned)__shfl_xor((int)tm, o));
    }
    __syncthreads();
    const int kl = threadIdx.x >> 2, mq = threadIdx.x & 3;       // output row k0 + kl, 16 consecutive m each
    const int k = k0 + kl;
    float tv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tv[i] = tile[mq * 16 + i][kl];
    if constexpr (SUMS) {
        float s4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) s4[g] = (tv[4 * g] + tv[4 * g + 1]) + (tv[4 * g + 2] + tv[4 * g + 3]);
        double sm = ((double)s4[0] + (double)s4[1]) + ((double)s4[2] + (double)s4[3]);
        sm += __shfl_xor(sm, 1);
        sm += __shfl_xor(sm, 2);
        if (mq == 0 && k < cols) part[(int64_t)blockIdx.y * cols + k] = sm;
    }
    if (k >= cols) return;
    float inv;
    const float s = x2_scale(__uint_as_float(TMAX ? tm : colmax[k]), &inv);
    unsigned short *hi = planes + (int64_t)k * 2 * mp + m0 + mq * 16, *lo = hi + mp;
    if (m0 + mq * 16 < mp) {                                      // (mp is a multiple of 16: whole 16-element pieces)
        unsigned short h[16], l[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x2_split(tv[i] * s, h[i], l[i]);
        uint4 *ph = reinterpret_cast<uint4 *>(hi), *pl = reinterpret_cast<uint4 *>(lo);
        ph[0] = *reinterpret_cast<uint4 *>(&h[0]); ph[1] = *reinterpret_cast<uint4 *>(&h[8]);
        pl[0] = *reinterpret_cast<uint4 *>(&l[0]); pl[1] = *reinterpret_cast<uint4 *>(&l[8]);
    }
    if (blockIdx.y == 0 && mq == 0) inv_scale[k] = inv;
}

// C[m][n] = epi(sum_z P[z][m][n]) in slot order (deterministic); bias / GELU / residual as the GEMM kernels order them
__global__ void x2_reduce_kernel(const float *P, int S, int M, int N, float *C, int64_t ldc, const float *bias, const float *residual,
                                 int64_t ldr, int epilogue) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += P[(int64_t)s * M * N + i];
    if ((epilogue & LA_EPI_BIAS) && bias) v += bias[n];
    if (epilogue & LA_EPI_GELU) v = la::gelu_erf(v);
    if ((epilogue & LA_EPI_RESIDUAL) && residual) {
        const float t = residual[(int64_t)m * ldr + n];
        v = (epilogue & LA_EPI_RES_GELU_GRAD) ? v * la::gelu_erf_grad(t) : v + t;
    }
    C[(int64_t)m * ldc + n] = v;
}

}  // namespace

extern "C" int la_split_f16x2_max(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale,
                                  int32_t act, uint32_t *tmax, void *stream_) {
    if (rows == 0) return LA_OK;
    LA_CHECK_ARG(act == 0 || act == 1, "split_f16x2: act is 0 (none) or 1 (GELU)");
    LA_CHECK_ARG(x && planes && inv_scale && rows > 0 && cols > 0, "split_f16x2: bad arguments");
    LA_CHECK_ARG(kp >= cols && kp % 8 == 0 && ldx >= cols && (uintptr_t)planes % 16 == 0, "split_f16x2: kp must be >= cols and a multiple of 8, planes 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    la::TimerScope ts("split_f16x2", stream, (double)rows * cols * 8.0);
    hipLaunchKernelGGL(split_rows_kernel, dim3(la::cdiv(rows, 4)), dim3(256), 0, stream, x, ldx, rows, cols,
                       reinterpret_cast<unsigned short *>(planes), kp, inv_scale, act, tmax);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_split_f16x2_act(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale,
                                  int32_t act, void *stream_) {
    return la_split_f16x2_max(x, ldx, rows, cols, planes, kp, inv_scale, act, nullptr, stream_);
}

extern "C" int la_split_f16x2(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale, void *stream_) {
    return la_split_f16x2_act(x, ldx, rows, cols, planes, kp, inv_scale, 0, stream_);
}

extern "C" int la_layernorm_f16x2(const float *x, int64_t ldx, int32_t rows, int32_t d, const float *gamma, const float *beta, void *planes,
                                  int64_t kp, float *inv_scale, void *stream_) {
    if (rows == 0) return LA_OK;
    LA_CHECK_ARG(x && gamma && beta && planes && inv_scale && rows > 0, "layernorm_f16x2: bad arguments");
    LA_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 4096 && ldx % 4 == 0 && ldx >= d && (uintptr_t)x % 16 == 0 && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0,
                 "layernorm_f16x2: d=%d unsupported (a multiple of 4, at most 4096) or misaligned rows", d);
    LA_CHECK_ARG(kp >= d && kp % 8 == 0 && (uintptr_t)planes % 16 == 0, "layernorm_f16x2: kp must be >= d and a multiple of 8, planes 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    la::TimerScope ts("split_f16x2", stream, (double)rows * d * 8.0);
    hipLaunchKernelGGL(ln_split_rows_kernel, dim3(la::cdiv(rows, 4)), dim3(256), 0, stream, x, ldx, rows, d, gamma, beta,
                       reinterpret_cast<unsigned short *>(planes), kp, inv_scale);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_colsum_f32(const float *in, int64_t ld, int32_t rows, int32_t cols, float *out, void *stream);

// colsum != NULL (act must be 0): also out[c] = sum_r x[r][c] -- the bias gradient next to the weight gradient whose operand this split
// makes, from the same pass over x that finds the column maxima
// tmax != NULL: the 64 words a plain split of the SAME operand (same act) left with la_split_f16x2_max -- the planes take one scale for
// the whole operand and the pass for column maxima is skipped; the column sums (colsum != NULL) then come out of the split kernel itself.
extern "C" int la_split_f16x2_t_tmax(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t,
                                     int32_t act, float *colsum, const uint32_t *tmax, void *stream_) {
    if (cols == 0) return LA_OK;
    LA_CHECK_ARG(act == 0 || act == 1, "split_f16x2_t: act is 0 (none) or 1 (GELU)");
    LA_CHECK_ARG(!colsum || act == 0, "split_f16x2_t: column sums are those of x itself (act = 0)");
    LA_CHECK_ARG(x && planes_t && inv_scale_t && rows > 0 && cols > 0, "split_f16x2_t: bad arguments");
    LA_CHECK_ARG(mp >= rows && mp % 16 == 0 && ldx >= cols && (uintptr_t)planes_t % 16 == 0, "split_f16x2_t: mp must be >= rows and a multiple of 16, planes 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    la::TimerScope ts("split_f16x2", stream, (double)rows * cols * 12.0);
    unsigned short *pl = reinterpret_cast<unsigned short *>(planes_t);
    const dim3 tgrid(la::cdiv(cols, 64), la::cdiv(mp, 64));
    if (tmax) {
        if (colsum) {
            double *part = static_cast<double *>(la::stream_scratch(stream, la::SCRATCH_COLSUM, sizeof(double) * (size_t)tgrid.y * cols));
            if (!part) { la::set_error("split_f16x2_t: scratch allocation failed"); return LA_EHIP; }
            hipLaunchKernelGGL((split_transposed_kernel<true, true>), tgrid, dim3(256), 0, stream, x, ldx, rows, cols, tmax, pl, mp, inv_scale_t, act, part);
            hipLaunchKernelGGL(colsum_blocks_kernel, dim3(la::cdiv(cols, 32)), dim3(256), 0, stream, part, (int)tgrid.y, cols, colsum);
        } else {
            hipLaunchKernelGGL((split_transposed_kernel<true, false>), tgrid, dim3(256), 0, stream, x, ldx, rows, cols, tmax, pl, mp, inv_scale_t, act,
                               (double *)nullptr);
        }
        LA_LAUNCH_CHECK();
        return LA_OK;
    }
    unsigned *colmax = static_cast<unsigned *>(la::stream_scratch(stream, la::SCRATCH_X2, (size_t)cols * sizeof(unsigned)));
    if (!colmax) { la::set_error("split_f16x2_t: scratch allocation failed"); return LA_EHIP; }
    LA_HIP(hipMemsetAsync(colmax, 0, (size_t)cols * sizeof(unsigned), stream));
    const int rpb = 512;
    if (cols % 4 == 0 && ldx % 4 == 0 && (uintptr_t)x % 16 == 0) {
        const dim3 grid(la::cdiv(cols, 256), la::cdiv(rows, 128));
        if (colsum) {
            double *part = static_cast<double *>(la::stream_scratch(stream, la::SCRATCH_COLSUM, sizeof(double) * (size_t)grid.y * cols));
            if (!part) { la::set_error("split_f16x2_t: scratch allocation failed"); return LA_EHIP; }
            hipLaunchKernelGGL(colmax4_kernel<true>, grid, dim3(256), 0, stream, x, ldx, rows, cols, 128, colmax, act, part);
            hipLaunchKernelGGL(colsum_blocks_kernel, dim3(la::cdiv(cols, 32)), dim3(256), 0, stream, part, (int)grid.y, cols, colsum);
        } else {
            hipLaunchKernelGGL(colmax4_kernel<false>, grid, dim3(256), 0, stream, x, ldx, rows, cols, 128, colmax, act, (double *)nullptr);
        }
    } else {
        hipLaunchKernelGGL(colmax_kernel, dim3(la::cdiv(cols, 64), la::cdiv(rows, rpb)), dim3(256), 0, stream, x, ldx, rows, cols, rpb, colmax, act);
        if (colsum) {
            const int rc = la_colsum_f32(x, ldx, rows, cols, colsum, stream_);
            if (rc != LA_OK) return rc;
        }
    }
    LA_LAUNCH_CHECK();
    hipLaunchKernelGGL((split_transposed_kernel<false, false>), tgrid, dim3(256), 0, stream, x, ldx, rows, cols, colmax, pl, mp, inv_scale_t, act,
                       (double *)nullptr);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_split_f16x2_t_colsum(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t,
                                       int32_t act, float *colsum, void *stream_) {
    return la_split_f16x2_t_tmax(x, ldx, rows, cols, planes_t, mp, inv_scale_t, act, colsum, nullptr, stream_);
}

extern "C" int la_split_f16x2_t_act(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t,
                                    int32_t act, void *stream_) {
    return la_split_f16x2_t_colsum(x, ldx, rows, cols, planes_t, mp, inv_scale_t, act, nullptr, stream_);
}

extern "C" int la_split_f16x2_t(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t, void *stream_) {
    return la_split_f16x2_t_act(x, ldx, rows, cols, planes_t, mp, inv_scale_t, 0, stream_);
}

extern "C" int la_gemm_f16x2(int32_t M, int32_t N, int32_t K, int32_t slots, const void *A, const float *sa, const void *W, const float *sw,
                             float *C, int64_t ldc, const float *bias, const float *residual, int64_t ldr, int32_t epilogue, void *stream_) {
    if (M == 0 || N == 0) return LA_OK;
    LA_CHECK_ARG(A && sa && W && sw && C, "gemm_f16x2: null pointer");
    LA_CHECK_ARG(M > 0 && N > 0 && K > 0 && slots >= 1, "gemm_f16x2: bad sizes");
    LA_CHECK_ARG((epilogue & ~(LA_EPI_BIAS | LA_EPI_GELU | LA_EPI_RESIDUAL | LA_EPI_RES_GELU_GRAD)) == 0,
                 "gemm_f16x2: epilogue takes BIAS, GELU, RESIDUAL, RES_GELU_GRAD only");
    LA_CHECK_ARG(!(epilogue & LA_EPI_RES_GELU_GRAD) || (epilogue & LA_EPI_RESIDUAL), "gemm_f16x2: RES_GELU_GRAD reads the residual operand (set RESIDUAL too)");
    LA_CHECK_ARG(!(epilogue & LA_EPI_BIAS) || bias, "gemm_f16x2: bias epilogue without pointer");
    LA_CHECK_ARG(!(epilogue & LA_EPI_RESIDUAL) || residual, "gemm_f16x2: residual epilogue without pointer");
    LA_CHECK_ARG((uintptr_t)A % 16 == 0 && (uintptr_t)W % 16 == 0, "gemm_f16x2: planes must be 16-byte aligned");
    const int Kc = K / slots;
    if (K % slots != 0 || Kc % 128 != 0 || Kc < 256 || N <= 128 || (int64_t)la::cdiv(M, 256) * la::cdiv(N, 256) * slots < 192) {
        la::set_error("gemm_f16x2: M=%d N=%d K=%d slots=%d outside the 256x256 kernel's domain (K / slots a multiple of 128 and >= 256, N > 128, "
                      ">= 192 tiles x slots)", M, N, K, slots);
        return LA_EUNSUPPORTED;
    }
    hipStream_t stream = (hipStream_t)stream_;
    float *out = C;
    int64_t out_ld = ldc, out_stride = 0;
    int epi = epilogue | LA_EPI_OUT_F32;
    if (slots > 1) {
        out = static_cast<float *>(la::stream_scratch(stream, la::SCRATCH_SPLITK, (size_t)slots * M * N * sizeof(float)));
        if (!out) { la::set_error("gemm_f16x2: split-K scratch allocation failed"); return LA_EHIP; }
        out_ld = N; out_stride = (int64_t)M * N;
        epi = LA_EPI_OUT_F32;                          // bias / activation / residual after the slots are summed
    }
    GemmParams p{M, N, Kc, A, (int64_t)2 * K, (int64_t)Kc, W, (int64_t)2 * K, (int64_t)Kc, out, out_ld, out_stride,
                 slots > 1 ? nullptr : bias, 0, slots > 1 ? nullptr : residual, ldr, 0, epi, 0, la::cdiv(N, BN), pick_group(3 * Kc, 2, la::cdiv(N, BN))};
    p.plane_a = K; p.plane_w = K;
    p.ln_stats = sa; p.ln_csum = sw;
    const int rc = launch_x2_f16(p, slots, stream);
    if (rc != LA_OK || slots == 1) return rc;
    const int64_t total = (int64_t)M * N;
    hipLaunchKernelGGL(x2_reduce_kernel, dim3((unsigned)la::cdiv(total, (int64_t)256)), dim3(256), 0, stream, out, slots, M, N, C, ldc, bias,
                       residual, ldr, epilogue);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
