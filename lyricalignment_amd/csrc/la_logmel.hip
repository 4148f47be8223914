// la_logmel.hip -- log-mel front end on the device (replaces whisper.audio.log_mel_spectrogram,
// which the reference runs on the CPU and then copies over: module/align_model.py:84).
//
//   audio [B][N] --reflect pad 200--> frames (overlapping 400-sample rows, hop 160)
//     --f32 MFMA GEMM with the windowed DFT matrix [cos | sin]--> re, im  --|.|^2--> power [201]
//     --f32 MFMA GEMM with the Slaney mel filter bank [80][201]--> mel
//     --log10(max(., 1e-10)), max over the WHOLE batch tensor, floor at max-8, (x+4)/4-->  [B][80][N/160]
// The STFT is a dense 400-point DFT expressed as a GEMM over the overlapping-row view (lda = 160):
// 31 GFLOP for 32 x 30 s, far below anything that matters, and it reuses the exact-f32 MFMA path.
// frame count = N / 160 (torch.stft(center=True) yields 1 + N/160 frames; upstream drops the last).
#include "la_common.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NMEL = 80;
constexpr int KPAD = 416;    // DFT length padded to the f32 GEMM K tile (32)
constexpr int NB_PAD = 208;  // bins padded so [cos | sin] is 2 x 208 rows
constexpr int PW_PAD = 224;  // power row pitch (K of the mel GEMM, multiple of 32)

// W[r][i]: r < 208 -> w[i] cos(2 pi r i / 400); r >= 208 -> w[i] sin(2 pi (r-208) i / 400); zero outside 201 bins / 400 taps
__global__ void build_dft_kernel(const float *window, float *W) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * NB_PAD * KPAD) return;
    const int r = idx / KPAD, i = idx % KPAD;
    const int k = r < NB_PAD ? r : r - NB_PAD;
    float v = 0.f;
    if (k < NBIN && i < NFFT) {
        const int ph = (int)(((long long)k * i) % NFFT);  // exact phase reduction
        const double a = 2.0 * (double)ph / (double)NFFT;  // in units of pi
        v = (float)((double)window[i] * (r < NB_PAD ? cospi(a) : sinpi(a)));
    }
    W[idx] = v;
}

// filters [80][201] -> [80][224] zero padded
__global__ void pad_filters_kernel(const float *f, float *fp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NMEL * PW_PAD) return;
    const int m = idx / PW_PAD, k = idx % PW_PAD;
    fp[idx] = k < NBIN ? f[m * NBIN + k] : 0.f;
}

// padded[b][i] = audio[b][reflect(i - 200)] for i < N + 400, zeros in the slack
__global__ void reflect_pad_kernel(const float *audio, int N, float *padded, int64_t pitch) {
    const int b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pitch) return;
    float v = 0.f;
    if (i < N + NFFT) {
        int64_t s = i - NFFT / 2;
        if (s < 0) s = -s;
        if (s >= N) s = 2 * (int64_t)(N - 1) - s;
        s = s < 0 ? 0 : s;  // only for N < 201, where torch.stft itself refuses the reflect pad
        v = audio[(int64_t)b * N + s];
    }
    padded[(int64_t)b * pitch + i] = v;
}

// spec [rows][416] (re | im) -> pw [rows][224]
__global__ void power_kernel(const float *spec, float *pw, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * PW_PAD) return;
    const int64_t r = idx / PW_PAD;
    const int k = (int)(idx % PW_PAD);
    float v = 0.f;
    if (k < NBIN) {
        const float re = spec[r * KPAD + k], im = spec[r * KPAD + NB_PAD + k];
        v = re * re + im * im;
    }
    pw[idx] = v;
}

// in place log10(max(x, 1e-10)) + per-block max
__global__ __launch_bounds__(256) void log_blockmax_kernel(float *x, int64_t n, float *blockmax) {
    __shared__ float red[4];
    float m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = log10f(fmaxf(x[i], 1e-10f));
        x[i] = v;
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// logspec [B][frames][80] -> mel [b][m][j] = (max(v, gmax - 8) + 4) / 4, tile transpose through LDS
__global__ __launch_bounds__(256) void finish_kernel(const float *logspec, const float *blockmax, int nblockmax, int frames,
                                                     float *mel, int64_t mbs, int64_t mrs) {
    __shared__ float tile[32][33];
    __shared__ float gmax_s;
    if (threadIdx.x < 64) {
        float m = -INFINITY;
        for (int i = threadIdx.x; i < nblockmax; i += 64) m = fmaxf(m, blockmax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (threadIdx.x == 0) gmax_s = m;
    }
    __syncthreads();
    const float floor_v = gmax_s - 8.0f;
    const int b = blockIdx.z, j0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int j = j0 + i, m = m0 + tx;
        float v = 0.f;
        if (j < frames && m < NMEL) v = logspec[((int64_t)b * frames + j) * NMEL + m];
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, j = j0 + tx;
        if (m < NMEL && j < frames) mel[(int64_t)b * mbs + (int64_t)m * mrs + j] = (fmaxf(tile[tx][i], floor_v) + 4.0f) / 4.0f;
    }
}

struct MelPlan {
    int frames;
    int64_t pitch;
    size_t off_pad, off_dft, off_filt, off_spec, off_pw, off_mel, off_bmax, total;
    int nblk;
};

MelPlan plan_mel(int batch, int n_samples) {
    MelPlan p;
    p.frames = n_samples / HOP;
    p.pitch = la::round_up((int64_t)n_samples + NFFT + 64, 64);
    const int64_t rows = (int64_t)batch * p.frames;
    p.nblk = (int)std::min<int64_t>(1024, std::max<int64_t>(1, la::cdiv(rows * NMEL, 256)));
    size_t o = 0;
    auto take = [&](int64_t bytes) { size_t r = o; o += (size_t)la::round_up(bytes, 256); return r; };
    p.off_pad = take((int64_t)batch * p.pitch * 4);
    p.off_dft = take((int64_t)2 * NB_PAD * KPAD * 4);
    p.off_filt = take((int64_t)NMEL * PW_PAD * 4);
    p.off_spec = take(rows * KPAD * 4);
    p.off_pw = take(rows * PW_PAD * 4);
    p.off_mel = take(rows * NMEL * 4);
    p.off_bmax = take((int64_t)p.nblk * 4);
    p.total = o;
    return p;
}

}  // namespace

extern "C" int la_logmel_workspace_bytes(int32_t batch, int32_t n_samples, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && n_samples >= HOP, "logmel_workspace_bytes: bad arguments");
    *bytes = plan_mel(batch, n_samples).total;
    return LA_OK;
}

extern "C" int la_logmel_f32(const float *audio, int32_t batch, int32_t n_samples, const float *mel_filters,
                             const float *window, float *mel, int64_t mel_batch_stride, int64_t mel_row_stride,
                             void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LA_CHECK_ARG(audio && mel_filters && window && mel && workspace, "logmel: null pointer");
    LA_CHECK_ARG(batch > 0 && n_samples > NFFT / 2, "logmel: need more than %d samples (reflect padding)", NFFT / 2);
    LA_CHECK_ARG((uintptr_t)workspace % 256 == 0, "logmel: workspace must be 256-byte aligned");
    const MelPlan p = plan_mel(batch, n_samples);
    LA_CHECK_ARG(p.frames >= 1 && mel_row_stride >= p.frames, "logmel: mel_row_stride < n_frames");
    LA_CHECK_ARG(workspace_bytes >= p.total, "logmel: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    float *padded = reinterpret_cast<float *>(ws + p.off_pad);
    float *dft = reinterpret_cast<float *>(ws + p.off_dft);
    float *filt = reinterpret_cast<float *>(ws + p.off_filt);
    float *spec = reinterpret_cast<float *>(ws + p.off_spec);
    float *pw = reinterpret_cast<float *>(ws + p.off_pw);
    float *ms = reinterpret_cast<float *>(ws + p.off_mel);
    float *bmax = reinterpret_cast<float *>(ws + p.off_bmax);
    const int64_t rows = (int64_t)batch * p.frames;

    la::TimerScope ts("logmel", stream);
    hipLaunchKernelGGL(build_dft_kernel, dim3(la::cdiv(2 * NB_PAD * KPAD, 256)), dim3(256), 0, stream, window, dft);
    hipLaunchKernelGGL(pad_filters_kernel, dim3(la::cdiv(NMEL * PW_PAD, 256)), dim3(256), 0, stream, mel_filters, filt);
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(la::cdiv(p.pitch, 256), batch), dim3(256), 0, stream, audio, n_samples, padded, p.pitch);
    LA_LAUNCH_CHECK();
    int rc = la::gemm_run(LA_F32, p.frames, 2 * NB_PAD, KPAD, batch, padded, HOP, p.pitch, dft, 0, spec, KPAD,
                          (int64_t)p.frames * KPAD, nullptr, 0, nullptr, 0, 0, LA_EPI_OUT_F32, stream);
    if (rc != LA_OK) return rc;
    hipLaunchKernelGGL(power_kernel, dim3(la::cdiv(rows * PW_PAD, 256)), dim3(256), 0, stream, spec, pw, rows);
    LA_LAUNCH_CHECK();
    rc = la::gemm_run(LA_F32, (int)rows, NMEL, PW_PAD, 1, pw, PW_PAD, 0, filt, 0, ms, NMEL, 0, nullptr, 0, nullptr, 0, 0,
                      LA_EPI_OUT_F32, stream);
    if (rc != LA_OK) return rc;
    hipLaunchKernelGGL(log_blockmax_kernel, dim3(p.nblk), dim3(256), 0, stream, ms, rows * NMEL, bmax);
    hipLaunchKernelGGL(finish_kernel, dim3(la::cdiv(p.frames, 32), la::cdiv(NMEL, 32), batch), dim3(256), 0, stream, ms, bmax,
                       p.nblk, p.frames, mel, mel_batch_stride, mel_row_stride);
    LA_LAUNCH_CHECK();
    return LA_OK;
}
