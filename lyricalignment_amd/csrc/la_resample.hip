// la_resample.hip -- rational-ratio polyphase FIR resampling to 16 kHz for the audio front end
// (utils/audio.py:3-20 of the reference calls librosa.load(sr=16000); SURVEY 8f "next" row 1).
//   y[n] = sum_k h[(n + skip) * down - k * up] * x[k]        (upfirdn with the centred, pre-padded low-pass h)
// One thread per output sample, ~len(h)/up taps each (55 for 44.1 kHz -> 16 kHz); h and the input window are
// L2-resident.  HBM-bound and tiny next to the encoder: 1.9 MB in, 1.9 MB out per 30 s clip.
#include "la_common.h"

namespace {
__global__ __launch_bounds__(256) void resample_kernel(const float *x, int64_t n_in, const float *h, int64_t h_len, int up,
                                                       int down, int64_t skip, float *y, int64_t n_out) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_out) return;
    const int64_t pos = (n + skip) * down;            // index into the (virtual) zero-stuffed convolution
    // taps: j = pos - k*up in [0, h_len)  <=>  k in [ceil((pos - h_len + 1)/up), floor(pos/up)]
    int64_t k_hi = pos / up;
    int64_t k_lo = (pos - h_len + 1 + up - 1);
    k_lo = k_lo <= 0 ? 0 : k_lo / up;
    if (k_hi > n_in - 1) k_hi = n_in - 1;
    double acc = 0.0;                                  // few dozen taps; double keeps the sum order-insensitive
    for (int64_t k = k_lo; k <= k_hi; ++k) acc += (double)h[pos - k * up] * (double)x[k];
    y[n] = (float)acc;
}
}  // namespace

extern "C" int la_resample_poly_f32(const float *x, int64_t n_in, const float *h, int64_t h_len, int32_t up, int32_t down,
                                    int64_t skip, float *y, int64_t n_out, void *stream_) {
    if (n_out == 0) return LA_OK;
    LA_CHECK_ARG(x && h && y && n_in > 0 && h_len > 0 && up > 0 && down > 0 && skip >= 0 && n_out > 0, "resample_poly: bad arguments");
    hipLaunchKernelGGL(resample_kernel, dim3(la::cdiv(n_out, 256)), dim3(256), 0, (hipStream_t)stream_, x, n_in, h, h_len, up,
                       down, skip, y, n_out);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
