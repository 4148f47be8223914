// la_logmel.hip -- log-mel front end on the device (replaces whisper.audio.log_mel_spectrogram,
// which the reference runs on the CPU and then copies over: module/align_model.py:84).
//
//   audio [B][N] --reflect pad 200--> frames (overlapping 400-sample rows, hop 160)
//     --windowed DFT [cos | sin] as an exact-f32 MFMA product--> re, im  --|.|^2--> power [201]
//     --Slaney mel filter bank [80][201], exact-f32 MFMA--> mel
//     --log10(max(., 1e-10)), max over the WHOLE batch tensor, floor at max-8, (x+4)/4-->  [B][80][N/160]
// frame count = N / 160 (torch.stft(center=True) yields 1 + N/160 frames; upstream drops the last).
//
// Round 4: ONE kernel per 64-frame tile does everything up to the log (logmel_tile_kernel):
//   * the tile's 10480 waveform samples (64 hops + the 240-sample overlap, reflect padding resolved on the way in) are
//     staged once in LDS; the 64 x 400 frame matrix is the overlapping row view of that span (row pitch 160 floats),
//     never materialised;
//   * a wave owns 16 frames and all 2 x 208 DFT columns: 26 accumulator tiles of v_mfma_f32_16x16x4_f32 (k-ordered fmaf
//     chains = the parity mode's arithmetic), the DFT matrix streams from L2 in fragment order (256 B per wave load);
//   * re^2 + im^2 in the accumulator registers, through LDS once (16 x 208 per wave) to become the K operand of the mel
//     product (5 accumulator tiles), log10 and the tile's maximum in registers, the tile leaves through LDS as whole
//     256-byte row segments of the [80][frames] output.
// Neither the spectrum [rows][416] nor the power [rows][224] of the first version exists in memory: HBM traffic is the
// waveform in (61 MB per 32 x 30 s), the log-mel out (31 MB) and one in-place pass for the whole-tensor floor
// (logmel_finish_kernel, 31 + 31 MB) = 1.7 x the 92 MB a log-mel must move; 2 launches (3 when the constants are built in
// the same call).  The constants (windowed DFT matrix in fragment order, padded filter bank) are built ONCE per
// (window, filters) pair by la_logmel_constants into a caller-owned buffer.
#include <algorithm>
#include <math.h>

#include "la_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NMEL = 80;
constexpr int FT = 64;                         // frames per workgroup tile (4 waves x 16)
constexpr int NCB = 13;                        // 16-bin column blocks per part: 208 >= 201 bins
constexpr int NBP = NCB * 16;                  // 208: padded bin count = K of the mel product (multiple of 4)
constexpr int NMB = NMEL / 16;                 // 5 blocks of 16 mel rows
constexpr int SPAN = (FT - 1) * HOP + NFFT;    // 10480 samples feed one tile
constexpr int PWP = 209;                       // power row pitch in floats (odd: 16 rows x 4 k land on distinct banks)
constexpr int OUTP = 68;                       // out tile row pitch
constexpr int DFT_FLOATS = 2 * NCB * NFFT * 16;    // [26 column blocks][400 taps][16 columns]
constexpr int FILT_FLOATS = NMB * NBP * 16;        // [5 mel blocks][208 bins][16 mel rows]
constexpr int LDS_MAIN = FT * PWP > SPAN ? FT * PWP : SPAN;   // the power tile reuses the waveform span
constexpr int LDS_FLOATS = LDS_MAIN + NMEL * OUTP + 4;

// consts[0 .. DFT_FLOATS): Wt[cb][k][c] = window[k] * cos | sin (2 pi bin k / 400), bin = (cb % 13) * 16 + c, cb < 13: cos,
//   zero for bin >= 201 -- so the B fragment of (column block cb, k-step s) is the 64 consecutive floats at (cb * 400 + 4 s) * 16;
// consts[DFT_FLOATS ..): Fp[mb][bin][r] = filters[mb * 16 + r][bin], zero for bin >= 201: the A fragment of (mel block, k-step).
__global__ void logmel_constants_kernel(const float *window, const float *filters, float *consts) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < DFT_FLOATS) {
        const int cb = idx / (NFFT * 16), k = (idx / 16) % NFFT, c = idx % 16;
        const int bin = (cb % NCB) * 16 + c;
        float v = 0.f;
        if (bin < NBIN) {
            const int ph = (int)(((long long)bin * k) % NFFT);   // exact phase reduction
            const double a = 2.0 * (double)ph / (double)NFFT;    // in units of pi
            v = (float)((double)window[k] * (cb < NCB ? cospi(a) : sinpi(a)));
        }
        consts[idx] = v;
    } else if (idx < DFT_FLOATS + FILT_FLOATS) {
        const int j = idx - DFT_FLOATS;
        const int mb = j / (NBP * 16), bin = (j / 16) % NBP, r = j % 16;
        consts[idx] = bin < NBIN ? filters[(mb * 16 + r) * NBIN + bin] : 0.f;
    }
}

// One 64-frame tile of one clip: waveform span -> log10 mel [80][64] (unfloored) into mel[b][m][j0 ..], the maximum of the
// tile's valid values into blockmax[b * gridDim.x + tile].
__global__ __launch_bounds__(256, 2) void logmel_tile_kernel(const float *__restrict__ audio, int N, int frames,
                                                             const float *__restrict__ consts, float *__restrict__ mel,
                                                             int64_t mbs, int64_t mrs, float *__restrict__ blockmax) {
    extern __shared__ float lds[];
    float *span = lds;                 // [SPAN] waveform, later pw[64][PWP]
    float *outt = lds + LDS_MAIN;      // [80][OUTP]
    float *red = outt + NMEL * OUTP;   // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y, j0 = blockIdx.x * FT;
    const float *a = audio + (int64_t)b * N;

    // ---- the tile's span of the reflect-padded waveform: padded[p] = audio[reflect(p - 200)], zero past N + 400 ----
    const int64_t p0 = (int64_t)j0 * HOP;
    for (int i = tid; i < SPAN; i += 256) {
        const int64_t p = p0 + i;
        float v = 0.f;
        if (p < (int64_t)N + NFFT) {
            int64_t s = p - NFFT / 2;
            if (s < 0) s = -s;
            if (s >= N) s = 2 * (int64_t)(N - 1) - s;
            s = s < 0 ? 0 : s;   // only for N < 201, where torch.stft itself refuses the reflect pad
            v = a[s];
        }
        span[i] = v;
    }
    __syncthreads();

    // ---- DFT: [16 frames of this wave] x [400 taps] x [26 column blocks] ----
    f32x4 acc[2 * NCB];
#pragma unroll
    for (int c = 0; c < 2 * NCB; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *xa = span + (16 * wave + lr) * HOP + lq;       // A: frame lr of the wave, tap 4 s + lq
    const float *wt = consts + lane;                            // B: (cb * 400 + 4 s) * 16 + lane
    float bcur[2 * NCB], bnext[2 * NCB];
#pragma unroll
    for (int c = 0; c < 2 * NCB; ++c) bcur[c] = wt[(int64_t)c * NFFT * 16];
    for (int s = 0; s < NFFT / 4; ++s) {
        const int sn = s + 1 < NFFT / 4 ? s + 1 : s;
#pragma unroll
        for (int c = 0; c < 2 * NCB; ++c) bnext[c] = wt[((int64_t)c * NFFT + 4 * sn) * 16];
        const float av = xa[4 * s];
#pragma unroll
        for (int c = 0; c < 2 * NCB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bcur[c], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 2 * NCB; ++c) bcur[c] = bnext[c];
    }
    __syncthreads();   // every wave is done with the waveform span: the power tile takes its place

    // ---- power = re^2 + im^2: lane holds bins cb * 16 + lr of frames 4 lq + r; rows of this wave only ----
    float *pw = span + (16 * wave) * PWP;
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(4 * lq + r) * PWP + c * 16 + lr] = acc[c][r] * acc[c][r] + acc[NCB + c][r] * acc[NCB + c][r];
    __syncthreads();

    // ---- mel^T[80][16 frames] = filters [80][208] x power^T [208][16 frames] ----
    f32x4 macc[NMB];
#pragma unroll
    for (int m = 0; m < NMB; ++m) macc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *fp = consts + DFT_FLOATS + lane;               // A: (mb * 208 + 4 s) * 16 + lane
    const float *pb = pw + lr * PWP + lq;                       // B: power[frame lr][bin 4 s + lq]
#pragma unroll 4
    for (int s = 0; s < NBP / 4; ++s) {
        const float bv = pb[4 * s];
#pragma unroll
        for (int m = 0; m < NMB; ++m) macc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[(m * NBP + 4 * s) * 16], bv, macc[m], 0, 0, 0);
    }

    // ---- log10, the tile's maximum over real frames, tile out through LDS as whole row segments ----
    const bool live = j0 + 16 * wave + lr < frames;
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < NMB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = log10f(fmaxf(macc[m][r], 1e-10f));
            if (live) mx = fmaxf(mx, v);
            outt[(m * 16 + 4 * lq + r) * OUTP + 16 * wave + lr] = v;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) blockmax[(int64_t)b * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int f = tid & 63;
    if (j0 + f < frames) {
        float *dst = mel + (int64_t)b * mbs + j0 + f;
        for (int m = tid >> 6; m < NMEL; m += 4) dst[(int64_t)m * mrs] = outt[m * OUTP + f];
    }
}

// mel[b][m][j] <- (max(v, gmax - 8) + 4) / 4 in place, gmax = the maximum over every tile of the call.
// One workgroup per (4 mel rows, clip).
__global__ __launch_bounds__(256) void logmel_finish_kernel(float *mel, int64_t mbs, int64_t mrs, int frames,
                                                            const float *blockmax, int nblockmax) {
    __shared__ float red[4];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nblockmax; i += 256) m = fmaxf(m, blockmax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const float floor_v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - 8.0f;
    for (int r = 0; r < 4; ++r) {
        float *row = mel + (int64_t)blockIdx.y * mbs + (int64_t)(blockIdx.x * 4 + r) * mrs;
        for (int j = threadIdx.x; j < frames; j += 256) row[j] = (fmaxf(row[j], floor_v) + 4.0f) / 4.0f;
    }
}

struct MelPlan {
    int frames, tiles;
    size_t off_consts, off_bmax, total;
};

MelPlan plan_mel(int batch, int n_samples) {
    MelPlan p;
    p.frames = n_samples / HOP;
    p.tiles = la::cdiv(p.frames, FT);
    size_t o = 0;
    auto take = [&](int64_t bytes) { size_t r = o; o += (size_t)la::round_up(bytes, 256); return r; };
    p.off_consts = take((int64_t)(DFT_FLOATS + FILT_FLOATS) * 4);
    p.off_bmax = take((int64_t)batch * p.tiles * 4);
    p.total = o;
    return p;
}

int run_logmel(const float *audio, int batch, int n_samples, const float *consts, float *mel, int64_t mbs, int64_t mrs,
               float *bmax, const MelPlan &p, hipStream_t stream) {
    static la::DeviceOnce attr_once;
    if (attr_once.pending()) {
        LA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(logmel_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LDS_FLOATS * 4));
        attr_once.mark();
    }
    hipLaunchKernelGGL(logmel_tile_kernel, dim3(p.tiles, batch), dim3(256), LDS_FLOATS * 4, stream, audio, n_samples, p.frames,
                       consts, mel, mbs, mrs, bmax);
    hipLaunchKernelGGL(logmel_finish_kernel, dim3(NMEL / 4, batch), dim3(256), 0, stream, mel, mbs, mrs, p.frames, bmax,
                       batch * p.tiles);
    LA_LAUNCH_CHECK();
    return LA_OK;
}

}  // namespace

extern "C" int la_logmel_constants_bytes(size_t *bytes) {
    LA_CHECK_ARG(bytes, "logmel_constants_bytes: null pointer");
    *bytes = (size_t)(DFT_FLOATS + FILT_FLOATS) * 4;
    return LA_OK;
}

extern "C" int la_logmel_constants(const float *mel_filters, const float *window, void *consts, size_t consts_bytes, void *stream_) {
    LA_CHECK_ARG(mel_filters && window && consts, "logmel_constants: null pointer");
    LA_CHECK_ARG(consts_bytes >= (size_t)(DFT_FLOATS + FILT_FLOATS) * 4 && (uintptr_t)consts % 16 == 0,
                 "logmel_constants: buffer too small or not 16-byte aligned");
    hipLaunchKernelGGL(logmel_constants_kernel, dim3(la::cdiv(DFT_FLOATS + FILT_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream_,
                       window, mel_filters, reinterpret_cast<float *>(consts));
    LA_LAUNCH_CHECK();
    return LA_OK;
}

extern "C" int la_logmel_workspace_bytes(int32_t batch, int32_t n_samples, size_t *bytes) {
    LA_CHECK_ARG(bytes && batch > 0 && n_samples >= HOP, "logmel_workspace_bytes: bad arguments");
    *bytes = plan_mel(batch, n_samples).total;
    return LA_OK;
}

static int logmel_checks(const float *audio, int32_t batch, int32_t n_samples, float *mel, int64_t mel_row_stride, void *workspace,
                         size_t workspace_bytes, const MelPlan &p) {
    LA_CHECK_ARG(audio && mel && workspace, "logmel: null pointer");
    LA_CHECK_ARG(batch > 0 && n_samples > NFFT / 2, "logmel: need more than %d samples (reflect padding)", NFFT / 2);
    LA_CHECK_ARG(batch <= 65535, "logmel: at most 65535 clips per call");
    LA_CHECK_ARG((uintptr_t)workspace % 256 == 0, "logmel: workspace must be 256-byte aligned");
    LA_CHECK_ARG(p.frames >= 1 && mel_row_stride >= p.frames, "logmel: mel_row_stride < n_frames");
    LA_CHECK_ARG(workspace_bytes >= p.total, "logmel: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    return LA_OK;
}

// constants built inside the call (3 launches)
extern "C" int la_logmel_f32(const float *audio, int32_t batch, int32_t n_samples, const float *mel_filters,
                             const float *window, float *mel, int64_t mel_batch_stride, int64_t mel_row_stride,
                             void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LA_CHECK_ARG(mel_filters && window, "logmel: null pointer");
    const MelPlan p = plan_mel(batch > 0 ? batch : 1, n_samples >= HOP ? n_samples : HOP);
    int rc = logmel_checks(audio, batch, n_samples, mel, mel_row_stride, workspace, workspace_bytes, p);
    if (rc != LA_OK) return rc;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    la::TimerScope ts("logmel", stream);
    rc = la_logmel_constants(mel_filters, window, ws + p.off_consts, (size_t)(DFT_FLOATS + FILT_FLOATS) * 4, stream_);
    if (rc != LA_OK) return rc;
    return run_logmel(audio, batch, n_samples, reinterpret_cast<const float *>(ws + p.off_consts), mel, mel_batch_stride,
                      mel_row_stride, reinterpret_cast<float *>(ws + p.off_bmax), p, stream);
}

// constants from la_logmel_constants (2 launches)
extern "C" int la_logmel_f32_prepared(const float *audio, int32_t batch, int32_t n_samples, const void *consts, float *mel,
                                      int64_t mel_batch_stride, int64_t mel_row_stride, void *workspace, size_t workspace_bytes,
                                      void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LA_CHECK_ARG(consts && (uintptr_t)consts % 16 == 0, "logmel: constants missing or misaligned");
    const MelPlan p = plan_mel(batch > 0 ? batch : 1, n_samples >= HOP ? n_samples : HOP);
    int rc = logmel_checks(audio, batch, n_samples, mel, mel_row_stride, workspace, workspace_bytes, p);
    if (rc != LA_OK) return rc;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    la::TimerScope ts("logmel", stream);
    return run_logmel(audio, batch, n_samples, reinterpret_cast<const float *>(consts), mel, mel_batch_stride, mel_row_stride,
                      reinterpret_cast<float *>(ws + p.off_bmax), p, stream);
}
