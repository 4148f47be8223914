// align_capi.cpp -- a consumer of liblyricalign_hip.so that is NOT Python: the whole alignment hot path through the C ABI's
// two model-level entry points (include/lyricalign.h):
//     mel [B][80][3000] f32  --la_encoder_forward-->  encoder rows  --la_align_head_forward-->  onset / offset frames
// which is what the reference does with whisper_model.embed_audio + align_rnn + perform_viterbi_ctc
// (module/align_model.py:91,107; utils/alignment.py:121-188).
//
//   align_capi <weights.bin> <input.bin> <output.bin>
//
// weights.bin / input.bin are written by lyricalignment_amd/capi_export.py (flat little-endian records, see read_*);
// output.bin receives  int32 B, Lmax; onset[B][Lmax], offset[B][Lmax], status[B] (int32); score[B] (float64).
// Build (done by lyricalignment_amd.build):  hipcc --offload-arch=gfx950 -O2 examples/align_capi.cpp -Iinclude \
//                                            -Llyricalignment_amd -llyricalign_hip -Wl,-rpath,'$ORIGIN/../../lyricalignment_amd'
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lyricalign.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define LA_CALL(x) do { int rc_ = (x); if (rc_ != LA_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, la_last_error()); return 3; } } while (0)

namespace {

struct Reader {
    FILE *f;
    bool ok = true;
    template <typename T> T scalar() { T v{}; if (fread(&v, sizeof(T), 1, f) != 1) ok = false; return v; }
    // one tensor record: int64 nbytes, then the bytes -> device memory (nullptr for an empty record)
    void *tensor() {
        const int64_t n = scalar<int64_t>();
        if (!ok || n < 0) { ok = false; return nullptr; }
        if (n == 0) return nullptr;
        std::vector<unsigned char> host((size_t)n);
        if (fread(host.data(), 1, (size_t)n, f) != (size_t)n) { ok = false; return nullptr; }
        void *d = nullptr;
        if (hipMalloc(&d, (size_t)n) != hipSuccess || hipMemcpy(d, host.data(), (size_t)n, hipMemcpyHostToDevice) != hipSuccess) { ok = false; return nullptr; }
        return d;
    }
};

}  // namespace

int main(int argc, char **argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s weights.bin input.bin output.bin\n", argv[0]); return 1; }
    if (!la_device_arch_ok()) { fprintf(stderr, "no gfx950 device: %s\n", la_last_error()); return 4; }
    Reader w{fopen(argv[1], "rb")};
    if (!w.f) { perror(argv[1]); return 1; }
    if (w.scalar<int32_t>() != 0x4C414331) { fprintf(stderr, "bad weights file\n"); return 1; }      // "LAC1"
    la_encoder_weights enc{};
    enc.dtype = w.scalar<int32_t>(); enc.d = w.scalar<int32_t>(); enc.n_head = w.scalar<int32_t>();
    enc.n_layer = w.scalar<int32_t>(); enc.n_mels = w.scalar<int32_t>();
    enc.conv1_w = w.tensor(); enc.conv1_b = (const float *)w.tensor();
    enc.conv2_w = w.tensor(); enc.conv2_b = (const float *)w.tensor();
    enc.pos = (const float *)w.tensor(); enc.lnp_g = (const float *)w.tensor(); enc.lnp_b = (const float *)w.tensor();
    std::vector<la_encoder_block> blocks((size_t)enc.n_layer);
    for (auto &b : blocks) {
        b.ln1_g = (const float *)w.tensor(); b.ln1_b = (const float *)w.tensor();
        b.wqkv = w.tensor(); b.bqkv = (const float *)w.tensor();
        b.wo = w.tensor(); b.bo = (const float *)w.tensor();
        b.ln2_g = (const float *)w.tensor(); b.ln2_b = (const float *)w.tensor();
        b.w1 = w.tensor(); b.b1 = (const float *)w.tensor();
        b.w2 = w.tensor(); b.b2 = (const float *)w.tensor();
        b.wqkv_ln = w.tensor(); b.cqkv = (const float *)w.tensor(); b.bqkv_ln = (const float *)w.tensor();
        b.w1_ln = w.tensor(); b.c1 = (const float *)w.tensor(); b.b1_ln = (const float *)w.tensor();
    }
    enc.blocks = blocks.data();
    la_head_weights head{};
    head.dtype = w.scalar<int32_t>(); head.hidden = w.scalar<int32_t>(); head.in_dim = w.scalar<int32_t>();
    head.vocab = w.scalar<int32_t>(); head.n_layers = w.scalar<int32_t>();
    for (int l = 0; l < 2; ++l) {
        head.w_ih[l] = w.tensor(); head.b_ih[l] = (const float *)w.tensor();
        head.w_hh[l] = w.tensor(); head.b_hh[l] = (const float *)w.tensor();
    }
    head.w_fc = w.tensor(); head.b_fc = (const float *)w.tensor();
    if (!w.ok) { fprintf(stderr, "truncated weights file\n"); return 1; }
    fclose(w.f);

    Reader in{fopen(argv[2], "rb")};
    if (!in.f) { perror(argv[2]); return 1; }
    const int32_t B = in.scalar<int32_t>(), frames = in.scalar<int32_t>(), Lmax = in.scalar<int32_t>(), variant = in.scalar<int32_t>();
    const float *mel = (const float *)in.tensor();                     // [B][n_mels][3000]
    const int32_t *labels = (const int32_t *)in.tensor();              // [B][Lmax]
    const int32_t *n_labels = (const int32_t *)in.tensor();            // [B]
    if (!in.ok || B <= 0) { fprintf(stderr, "bad input file\n"); return 1; }
    fclose(in.f);

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    const size_t es = (enc.dtype & 0xff) == LA_F32 ? 4 : 2;   // (LA_Q_LOG2 may ride on the weights' dtype)
    size_t ws_e = 0, ws_h = 0;
    LA_CALL(la_encoder_workspace_bytes(&enc, B, &ws_e));
    LA_CALL(la_align_head_workspace_bytes(&head, B, frames, Lmax, &ws_h));
    void *feats, *wse, *wsh;
    int32_t *onset, *offset, *status, *flag;
    double *score;
    HIP_OK(hipMalloc(&feats, (size_t)B * 1500 * enc.d * es));
    HIP_OK(hipMalloc(&wse, ws_e)); HIP_OK(hipMalloc(&wsh, ws_h));
    HIP_OK(hipMalloc(&onset, (size_t)B * Lmax * 4)); HIP_OK(hipMalloc(&offset, (size_t)B * Lmax * 4));
    HIP_OK(hipMalloc(&status, (size_t)B * 4)); HIP_OK(hipMalloc(&score, (size_t)B * 8)); HIP_OK(hipMalloc(&flag, 4));
    HIP_OK(hipMemsetAsync(flag, 0, 4, stream));
    LA_CALL(la_encoder_forward(&enc, mel, (int64_t)enc.n_mels * 3000, 3000, B, feats, enc.d, enc.dtype & 0xff, wse, ws_e, stream));
    LA_CALL(la_align_head_forward(&head, feats, enc.d, 1500, B, frames, variant, labels, Lmax, n_labels, Lmax, onset, offset, Lmax, score,
                                  status, nullptr, wsh, ws_h, flag, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<int32_t> h_on((size_t)B * Lmax), h_off((size_t)B * Lmax), h_st((size_t)B);
    std::vector<double> h_sc((size_t)B);
    int32_t h_flag = 0;
    HIP_OK(hipMemcpy(h_on.data(), onset, h_on.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_off.data(), offset, h_off.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_st.data(), status, h_st.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_sc.data(), score, h_sc.size() * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&h_flag, flag, 4, hipMemcpyDeviceToHost));
    if (h_flag) { fprintf(stderr, "persistent GRU kernel: a bounded wait timed out\n"); return 5; }
    FILE *out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 1; }
    fwrite(&B, 4, 1, out); fwrite(&Lmax, 4, 1, out);
    fwrite(h_on.data(), 4, h_on.size(), out); fwrite(h_off.data(), 4, h_off.size(), out);
    fwrite(h_st.data(), 4, h_st.size(), out); fwrite(h_sc.data(), 8, h_sc.size(), out);
    fclose(out);
    printf("aligned %d clips x %d frames, up to %d labels: first onset / offset frames %d / %d, status %d\n", B, frames, Lmax, h_on[0], h_off[0], h_st[0]);
    return 0;
}
