// la_model.cpp -- model-level entry points: the kernel SEQUENCES of the hot path behind one C call each, so that a consumer
// that is not Python does not have to re-implement lyricalignment_amd/engine.py:
//   la_encoder_forward     whisper_model.embed_audio(mel)          module/align_model.py:91,101,112,137
//   la_align_head_forward  align_rnn(embed) -> emission prep -> perform_viterbi(_ctc) -> onset / offset frames
//                          module/align_model.py:35-38,107,115; utils/alignment.py:13-71,121-188
// Host code only: it enqueues the op-level kernels of this library (la_gemm, la_attention, la_gru_layer, ...) on the
// caller's stream in the order engine.AlignEngine.encode / head_hidden / emissions do, out of a caller-provided workspace.
#include "la_common.h"

namespace {

constexpr int N_FRAMES = 3000, N_CTX = 1500, C_PAD = 128;

struct Carve {
    unsigned char *base;
    size_t off = 0;
    explicit Carve(void *p) : base(static_cast<unsigned char *>(p)) {}
    void *take(size_t bytes) {
        void *p = base ? base + off : nullptr;
        off += (size_t)la::round_up((int64_t)bytes, 256);
        return p;
    }
};

size_t esize(int dtype) { return dtype == LA_F32 ? 4 : 2; }

struct EncBufs {
    void *rows0, *y1, *h, *qkv, *att, *u;
    float *x, *stats;
    // float32 on the f16x2 products: operand planes of the d-wide operands (LN(x), attention output; one buffer, used in turn) and of the
    // MLP's hidden operand, their per-row inverse scales, the attention's own workspace
    void *pl_d = nullptr, *pl_u = nullptr, *attn_ws = nullptr;
    float *inv = nullptr;
    size_t attn_ws_bytes = 0;
    size_t total;
};

// Does the float32 encoder run its blocks on the f16 matrix pipe (la_gemm_f16x2 + la_layernorm_f16x2 + la_attention_lse_f16x2)?  The blocks
// carry the weight planes, the option is on, and every Linear of a block lies in the 256 x 256 kernel's domain: >= 192 tiles for the narrowest
// (N = d), K = d a multiple of 128 and >= 256.
bool encoder_x2(const la_encoder_weights *w, int batch) {
    if ((w->dtype & 0xff) != LA_F32 || w->n_layer < 1 || !la::opts().x2_inference) return false;
    const int d = w->d;
    if (d % 128 != 0 || d < 256 || d > 1024 * 4) return false;
    for (int l = 0; l < w->n_layer; ++l) {
        const la_encoder_block &k = w->blocks[l];
        if (!(k.wqkv_x2 && k.wqkv_x2s && k.wo_x2 && k.wo_x2s && k.w1_x2 && k.w1_x2s && k.w2_x2 && k.w2_x2s)) return false;
    }
    return la::cdiv((int64_t)batch * N_CTX, 256) * la::cdiv(d, 256) >= 192;
}

EncBufs carve_encoder(void *ws, int dtype, int batch, int d, bool x2 = false, int n_head = 0) {
    Carve c(ws);
    const size_t es = esize(dtype), M = (size_t)batch * N_CTX;
    EncBufs b;
    // the conv stem's buffers (channels-last mel rows, conv1 output) are dead once conv2 has run, long before the first MLP-up
    // GEMM writes `u`: they live inside u's region (u = 4 d per row of M = 1500 B rows >= 3002 B rows x (128 + d) for d >= 128)
    const size_t stem = (size_t)la::round_up((int64_t)((size_t)batch * (N_FRAMES + 2) * C_PAD * es), 256) + (size_t)batch * (N_FRAMES + 2) * d * es;
    const size_t u_bytes = M * 4 * d * es;
    b.u = c.take(u_bytes > stem ? u_bytes : stem);
    b.rows0 = b.u;
    b.y1 = b.u ? static_cast<unsigned char *>(b.u) + (size_t)la::round_up((int64_t)((size_t)batch * (N_FRAMES + 2) * C_PAD * es), 256) : nullptr;
    b.x = static_cast<float *>(c.take(M * d * 4));
    b.h = c.take(M * d * es);
    b.qkv = c.take(M * 3 * d * es);
    b.att = c.take(M * d * es);
    b.stats = static_cast<float *>(c.take(M * 2 * 4));
    if (x2) {
        b.pl_d = c.take(M * 2 * d * 2);
        b.pl_u = c.take(M * 2 * 4 * d * 2);
        b.inv = static_cast<float *>(c.take(M * 4));
        la_attention_f16x2_workspace_bytes(batch, N_CTX, N_CTX, n_head, &b.attn_ws_bytes);
        b.attn_ws = c.take(b.attn_ws_bytes);
    }
    b.total = c.off;
    return b;
}

bool encoder_fused_ln(const la_encoder_weights *w, int batch) {
    if ((w->dtype & 0xff) == LA_F32 || w->n_layer < 1 || !w->blocks[0].wqkv_ln || w->d <= 128) return false;
    if (!la::opts().ln_fusion) return false;
    const int64_t M = (int64_t)batch * N_CTX;
    // every GEMM of a block (and the batched conv2) must run on the 256x256 kernel the fold is built into: >= 192 tiles
    return la::cdiv(M, 256) * la::cdiv(w->d, 256) >= 192 && (int64_t)la::cdiv(N_CTX, 256) * la::cdiv(w->d, 256) * batch >= 192;
}

#define LA_TRY(expr)                   \
    do {                               \
        const int rc_ = (expr);        \
        if (rc_ != LA_OK) return rc_;  \
    } while (0)

}  // namespace

extern "C" int la_encoder_workspace_bytes(const la_encoder_weights *w, int32_t batch, size_t *bytes) {
    LA_CHECK_ARG(w && bytes && batch > 0 && w->d > 0, "encoder_workspace_bytes: bad arguments");
    *bytes = carve_encoder(nullptr, w->dtype & 0xff, batch, w->d, encoder_x2(w, batch), w->n_head).total;
    return LA_OK;
}

extern "C" int la_encoder_forward(const la_encoder_weights *w, const float *mel, int64_t mel_batch_stride, int64_t mel_row_stride,
                                  int32_t batch, void *out, int64_t ld_out, int32_t out_dtype, void *workspace,
                                  size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return LA_OK;
    LA_CHECK_ARG(w && mel && out && workspace && batch > 0, "encoder_forward: null pointer / bad batch");
    LA_CHECK_ARG(((w->dtype & 0xff) == LA_F32 && !(w->dtype & LA_Q_LOG2)) || (w->dtype & ~LA_Q_LOG2) == LA_BF16 || (w->dtype & ~LA_Q_LOG2) == LA_F16,
                 "encoder_forward: bad compute dtype");
    LA_CHECK_ARG(w->d > 0 && w->d % 64 == 0 && w->n_head * 64 == w->d, "encoder_forward: kernels are built for head_dim 64 (d = %d, heads = %d)", w->d, w->n_head);
    LA_CHECK_ARG(w->n_mels > 0 && w->n_mels <= C_PAD && w->n_layer >= 0 && (w->n_layer == 0 || w->blocks), "encoder_forward: bad dimensions");
    LA_CHECK_ARG((uintptr_t)workspace % 256 == 0, "encoder_forward: workspace must be 256-byte aligned");
    const int dt = w->dtype & 0xff, dt_attn = w->dtype, d = w->d, M = batch * N_CTX;   // (LA_Q_LOG2 rides on the attention calls' dtype)
    const bool x2 = encoder_x2(w, batch);
    const EncBufs b = carve_encoder(workspace, dt, batch, d, x2, w->n_head);
    LA_CHECK_ARG(workspace_bytes >= b.total, "encoder_forward: workspace too small (%zu < %zu)", workspace_bytes, b.total);
    const size_t es = esize(dt);
    const int out_f32 = dt == LA_F32 ? 0 : LA_EPI_OUT_F32;       // the residual stream is f32 in every mode

    // conv stem as GEMMs over overlapping row views of channels-last, zero-bordered activations (engine.encode)
    LA_TRY(la_mel_to_rows(mel, mel_batch_stride, mel_row_stride, batch, w->n_mels, N_FRAMES, b.rows0, C_PAD, dt, stream));
    const size_t clip1 = (size_t)(N_FRAMES + 2) * d * es;
    LA_HIP(hipMemset2DAsync(b.y1, clip1, 0, (size_t)d * es, batch, stream));                                        // row 0 of every clip
    LA_HIP(hipMemset2DAsync(static_cast<unsigned char *>(b.y1) + (size_t)(N_FRAMES + 1) * d * es, clip1, 0, (size_t)d * es, batch, stream));
    LA_TRY(la_gemm(dt, N_FRAMES, d, 3 * C_PAD, batch, b.rows0, C_PAD, (int64_t)(N_FRAMES + 2) * C_PAD, w->conv1_w,
                   static_cast<unsigned char *>(b.y1) + (size_t)d * es, d, (int64_t)(N_FRAMES + 2) * d, w->conv1_b, nullptr, 0, 0,
                   LA_EPI_BIAS | LA_EPI_GELU, stream));
    const bool fused = encoder_fused_ln(w, batch);
    const int epi2 = LA_EPI_BIAS | LA_EPI_GELU | LA_EPI_RESIDUAL | out_f32;
    // With the LayerNorm fold the residual stream is kept SPLIT (la_gemm_split: hi = b.h, the next GEMM's raw operand, + one lo byte
    // per element in the first quarter of b.x) instead of f32 with a 16-bit copy beside it: the residual GEMMs' epilogues are bound
    // by the stream's bytes.  Option resid_split = 0 (LA_RESID_SPLIT) keeps the f32 stream: the A/B partner.
    const bool split = fused && la::opts().resid_split;
    unsigned char *lo = reinterpret_cast<unsigned char *>(b.x);
    // Row statistics of the folded LayerNorms: la_row_stats16 over the stream's hi rows.  (Experiment build, LA_LN_STATS=loop, bfloat16:
    // the consumer GEMM's own main loop takes them from the A fragments it multiplies -- la_gemm_fused_ln with ln_stats = NULL, K = d a
    // multiple of 128.  Measured slower: the extra vector instructions cost the loop more than the pass they remove.)
    const char *stats_env = la::dev_env("LA_LN_STATS");
    const char *dbg_env = la::dev_env("LA_PP_DBG");
    const bool stats_in_loop = fused && dt == LA_BF16 && d % 128 == 0 && d >= 256 && stats_env && strcmp(stats_env, "loop") == 0 &&
                               !(dbg_env && (atoi(dbg_env) == 99 || atoi(dbg_env) == 73));
    const float *ln_stats = stats_in_loop ? nullptr : b.stats;
    if (split) {
        LA_TRY(la_gemm_split(dt, N_CTX, d, 3 * d, batch, b.y1, 2 * d, (int64_t)(N_FRAMES + 2) * d, w->conv2_w, b.h, lo, d, (int64_t)N_CTX * d,
                             w->conv2_b, w->pos, d, 0, LA_EPI_BIAS | LA_EPI_GELU | LA_EPI_RESIDUAL, nullptr, stream));
        if (!stats_in_loop) LA_TRY(la_row_stats16(dt, b.h, d, M, d, 1e-5f, b.stats, stream));
    } else if (fused) {
        LA_TRY(la_gemm_fused_ln(dt, N_CTX, d, 3 * d, batch, b.y1, 2 * d, (int64_t)(N_FRAMES + 2) * d, w->conv2_w, b.x, d, (int64_t)N_CTX * d,
                                w->conv2_b, w->pos, d, 0, epi2, b.h, d, (int64_t)N_CTX * d, nullptr, nullptr, nullptr, stream));
        if (!stats_in_loop) LA_TRY(la_row_stats16(dt, b.h, d, M, d, 1e-5f, b.stats, stream));
    } else {
        LA_TRY(la_gemm(dt, N_CTX, d, 3 * d, batch, b.y1, 2 * d, (int64_t)(N_FRAMES + 2) * d, w->conv2_w, b.x, d, (int64_t)N_CTX * d,
                       w->conv2_b, w->pos, d, 0, epi2, stream));
    }
    const int epi_res = LA_EPI_BIAS | LA_EPI_RESIDUAL | out_f32;
    for (int l = 0; l < w->n_layer; ++l) {
        const la_encoder_block &k = w->blocks[l];
        if (fused) {
            LA_CHECK_ARG(k.wqkv_ln && k.cqkv && k.bqkv_ln && k.w1_ln && k.c1 && k.b1_ln, "encoder_forward: block %d lacks the LayerNorm-folded weights", l);
            LA_TRY(la_gemm_fused_ln(dt, M, 3 * d, d, 1, b.h, d, 0, k.wqkv_ln, b.qkv, 3 * d, 0, k.bqkv_ln, nullptr, 0, 0, LA_EPI_BIAS, nullptr, 0, 0,
                                    ln_stats, k.cqkv, nullptr, stream));
            LA_TRY(la_attention(dt_attn, b.qkv, 3 * d, b.att, d, batch, N_CTX, w->n_head, stream));
            if (split) LA_TRY(la_gemm_split(dt, M, d, d, 1, b.att, d, 0, k.wo, b.h, lo, d, 0, k.bo, nullptr, 0, 0, LA_EPI_BIAS | LA_EPI_RESIDUAL, nullptr, stream));
            else LA_TRY(la_gemm_fused_ln(dt, M, d, d, 1, b.att, d, 0, k.wo, b.x, d, 0, k.bo, b.x, d, 0, epi_res, b.h, d, 0, nullptr, nullptr, nullptr, stream));
            if (!stats_in_loop) LA_TRY(la_row_stats16(dt, b.h, d, M, d, 1e-5f, b.stats, stream));
            LA_TRY(la_gemm_fused_ln(dt, M, 4 * d, d, 1, b.h, d, 0, k.w1_ln, b.u, 4 * d, 0, k.b1_ln, nullptr, 0, 0, LA_EPI_BIAS | LA_EPI_GELU, nullptr, 0, 0,
                                    ln_stats, k.c1, nullptr, stream));
            if (split) LA_TRY(la_gemm_split(dt, M, d, 4 * d, 1, b.u, 4 * d, 0, k.w2, b.h, lo, d, 0, k.b2, nullptr, 0, 0, LA_EPI_BIAS | LA_EPI_RESIDUAL, nullptr, stream));
            else LA_TRY(la_gemm_fused_ln(dt, M, d, 4 * d, 1, b.u, 4 * d, 0, k.w2, b.x, d, 0, k.b2, b.x, d, 0, epi_res, b.h, d, 0, nullptr, nullptr, nullptr, stream));
            if (!stats_in_loop) LA_TRY(la_row_stats16(dt, b.h, d, M, d, 1e-5f, b.stats, stream));
        } else if (x2) {
            // float32 on the f16 matrix pipe at float32 accuracy (csrc/la_f32x2.hip): every operand as two IEEE-half planes with a power-of-two
            // scale per row, three f16 products per Linear in one pass of the 256 x 256 kernel.  LayerNorm leaves as the next Linear's planes
            // (no float32 copy), the MLP's gelu(u) is applied inside its operand split, the residual adds stay in the float32 stream b.x.
            float *qkv = static_cast<float *>(b.qkv), *att = static_cast<float *>(b.att), *u = static_cast<float *>(b.u);
            LA_TRY(la_layernorm_f16x2(b.x, d, M, d, k.ln1_g, k.ln1_b, b.pl_d, d, b.inv, stream));
            LA_TRY(la_gemm_f16x2(M, 3 * d, d, 1, b.pl_d, b.inv, k.wqkv_x2, k.wqkv_x2s, qkv, 3 * d, k.bqkv, nullptr, 0, LA_EPI_BIAS, stream));
            LA_TRY(la_attention_lse_f16x2(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, d, batch, N_CTX, N_CTX, w->n_head, 0, nullptr, b.attn_ws,
                                          b.attn_ws_bytes, stream));
            LA_TRY(la_split_f16x2(att, d, M, d, b.pl_d, d, b.inv, stream));
            LA_TRY(la_gemm_f16x2(M, d, d, 1, b.pl_d, b.inv, k.wo_x2, k.wo_x2s, b.x, d, k.bo, b.x, d, LA_EPI_BIAS | LA_EPI_RESIDUAL, stream));
            LA_TRY(la_layernorm_f16x2(b.x, d, M, d, k.ln2_g, k.ln2_b, b.pl_d, d, b.inv, stream));
            LA_TRY(la_gemm_f16x2(M, 4 * d, d, 1, b.pl_d, b.inv, k.w1_x2, k.w1_x2s, u, 4 * d, k.b1, nullptr, 0, LA_EPI_BIAS, stream));
            LA_TRY(la_split_f16x2_act(u, 4 * d, M, 4 * d, b.pl_u, 4 * d, b.inv, 1 /* exact-erf GELU */, stream));
            LA_TRY(la_gemm_f16x2(M, d, 4 * d, 1, b.pl_u, b.inv, k.w2_x2, k.w2_x2s, b.x, d, k.b2, b.x, d, LA_EPI_BIAS | LA_EPI_RESIDUAL, stream));
        } else {
            LA_TRY(la_layernorm(b.x, d, M, d, k.ln1_g, k.ln1_b, b.h, d, dt, stream));
            LA_TRY(la_gemm(dt, M, 3 * d, d, 1, b.h, d, 0, k.wqkv, b.qkv, 3 * d, 0, k.bqkv, nullptr, 0, 0, LA_EPI_BIAS, stream));
            LA_TRY(la_attention(dt_attn, b.qkv, 3 * d, b.att, d, batch, N_CTX, w->n_head, stream));
            LA_TRY(la_gemm(dt, M, d, d, 1, b.att, d, 0, k.wo, b.x, d, 0, k.bo, b.x, d, 0, epi_res, stream));
            LA_TRY(la_layernorm(b.x, d, M, d, k.ln2_g, k.ln2_b, b.h, d, dt, stream));
            LA_TRY(la_gemm(dt, M, 4 * d, d, 1, b.h, d, 0, k.w1, b.u, 4 * d, 0, k.b1, nullptr, 0, 0, LA_EPI_BIAS | LA_EPI_GELU, stream));
            LA_TRY(la_gemm(dt, M, d, 4 * d, 1, b.u, 4 * d, 0, k.w2, b.x, d, 0, k.b2, b.x, d, 0, epi_res, stream));
        }
    }
    if (split) return la_layernorm_split(dt, b.h, lo, d, M, d, w->lnp_g, w->lnp_b, out, ld_out, out_dtype, stream);
    return la_layernorm(b.x, d, M, d, w->lnp_g, w->lnp_b, out, ld_out, out_dtype, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
namespace {

int head_clip_cap(const la_head_weights *w, int frames) {
    const int nsplit = w->hidden / (16 * (w->dtype == LA_F32 ? 2 : 4));
    const int64_t by_cus = (224 / (2 * nsplit)) * 16;
    const int64_t by_desc = (int64_t)2147483647 / ((int64_t)frames * 2 * w->hidden * (int64_t)esize(w->dtype));
    int64_t cap = by_cus < by_desc ? by_cus : by_desc;
    if (cap > 256) cap = 256;
    const int force = la::opts().head_clip_cap;               // option head_clip_cap: exercise the slicing with a handful of clips
    if (force > 0 && force < cap) cap = force;
    return cap < 1 ? 1 : (int)cap;
}

struct HeadBufs {
    float *gi, *em;
    void *gru[2], *act, *gru_ws, *fc_ws, *vit_ws;
    int32_t *n_frames;
    // float32 on the f16x2 products: the planes of an input projection's operand rows (encoder output / first layer's output) + inverse scales
    void *planes = nullptr;
    float *inv = nullptr;
    size_t gru_ws_bytes, fc_ws_bytes, vit_ws_bytes, total;
};

// float32 head on the f16 matrix pipe (input projections, recurrence, output Linear's normaliser): the weights carry the planes and the option is on
bool head_x2(const la_head_weights *w) {
    return w->dtype == LA_F32 && la::opts().x2_inference && w->w_ih_x2[0] && w->w_ih_x2s[0] && w->w_ih_x2[1] && w->w_ih_x2s[1] && w->w_fc_x2 && w->w_fc_x2s;
}
// one input projection [rows][K] x [6H][K]^T in la_gemm_f16x2's domain?
bool proj_x2_ok(int64_t rows, int K, int H) { return K % 128 == 0 && K >= 256 && la::cdiv(rows, 256) * la::cdiv(6 * H, 256) >= 192; }

int carve_head(void *ws, const la_head_weights *w, int batch, int frames, int max_labels, HeadBufs *b) {
    Carve c(ws);
    const size_t es = esize(w->dtype);
    // every buffer is sized for ONE slice of clips (bb): GRU, FC and DP run slice by slice, so a 4608-clip batch (BASELINE
    // configs[3]) holds 256 clips' worth of Mish(GRU) and emissions here, not the whole batch's (10.6 GB of `act` in round 2)
    const int H = w->hidden, cap = head_clip_cap(w, frames), bb = batch < cap ? batch : cap;
    b->gi = static_cast<float *>(c.take((size_t)bb * frames * 6 * H * 4));
    b->gru[0] = c.take((size_t)bb * frames * 2 * H * es);
    b->gru[1] = c.take((size_t)bb * frames * 2 * H * es);
    b->act = c.take((size_t)bb * frames * 2 * H * es);
    b->em = static_cast<float *>(c.take((size_t)bb * frames * (max_labels + 1) * 4));
    b->n_frames = static_cast<int32_t *>(c.take((size_t)bb * 4));
    LA_TRY(la_gru_workspace_bytes(bb, frames, H, &b->gru_ws_bytes));
    b->gru_ws = c.take(b->gru_ws_bytes);
    if (head_x2(w)) {
        const int kmax = w->in_dim > 2 * H ? w->in_dim : 2 * H;
        b->planes = c.take((size_t)bb * frames * 2 * kmax * 2);
        b->inv = static_cast<float *>(c.take((size_t)bb * frames * 4));
        LA_TRY(la_fc_emissions_x2_workspace_bytes(bb, frames, 2 * H, w->vocab, max_labels, &b->fc_ws_bytes));
    } else {
        LA_TRY(la_fc_emissions_workspace_bytes(w->dtype, bb, frames, 2 * H, w->vocab, max_labels, &b->fc_ws_bytes));
    }
    b->fc_ws = c.take(b->fc_ws_bytes);
    LA_TRY(la_viterbi_workspace_bytes(bb, frames, max_labels, &b->vit_ws_bytes));
    b->vit_ws = c.take(b->vit_ws_bytes > 16 ? b->vit_ws_bytes : 16);
    b->total = c.off;
    return LA_OK;
}

}  // namespace

extern "C" int la_align_head_workspace_bytes(const la_head_weights *w, int32_t batch, int32_t frames, int32_t max_labels, size_t *bytes) {
    LA_CHECK_ARG(w && bytes && batch > 0 && frames > 0 && max_labels > 0 && w->hidden > 0 && w->hidden % 64 == 0, "align_head_workspace_bytes: bad arguments");
    HeadBufs b;
    LA_TRY(carve_head(nullptr, w, batch, frames, max_labels, &b));
    *bytes = b.total;
    return LA_OK;
}

extern "C" int la_align_head_forward(const la_head_weights *w, const void *feats, int64_t ld_feats, int64_t clip_stride_rows,
                                     int32_t batch, int32_t frames, int32_t variant, const int32_t *labels, int32_t labels_stride,
                                     const int32_t *n_labels, int32_t max_labels, int32_t *onset, int32_t *offset, int32_t out_stride,
                                     double *final_score, int32_t *status, float *emissions_out, void *workspace,
                                     size_t workspace_bytes, int32_t *timeout_flag, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return LA_OK;
    LA_CHECK_ARG(w && feats && labels && n_labels && onset && offset && final_score && status && workspace, "align_head_forward: null pointer");
    LA_CHECK_ARG(batch > 0 && frames > 0 && max_labels > 0 && w->n_layers == 2 && w->hidden % 64 == 0, "align_head_forward: bad sizes (2 GRU layers expected)");
    LA_CHECK_ARG((uintptr_t)workspace % 256 == 0, "align_head_forward: workspace must be 256-byte aligned");
    HeadBufs b;
    LA_TRY(carve_head(workspace, w, batch, frames, max_labels, &b));
    LA_CHECK_ARG(workspace_bytes >= b.total, "align_head_forward: workspace too small (%zu < %zu)", workspace_bytes, b.total);
    const int dt = w->dtype, H = w->hidden;
    const size_t es = esize(dt);
    const int out_f32 = dt == LA_F32 ? 0 : LA_EPI_OUT_F32;
    const int cap = head_clip_cap(w, frames);
    const bool x2 = head_x2(w);
    LA_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b.n_frames), frames, batch < cap ? batch : cap, stream));
    const int64_t em_clip = (int64_t)frames * (max_labels + 1);
    for (int b0 = 0; b0 < batch; b0 += cap) {        // the persistent recurrence takes one launch set of clips at a time
        const int nb = batch - b0 < cap ? batch - b0 : cap;
        const unsigned char *x = static_cast<const unsigned char *>(feats) + (size_t)b0 * clip_stride_rows * ld_feats * es;
        int64_t lda = ld_feats, stride_a = clip_stride_rows * ld_feats;
        int in_dim = w->in_dim;
        for (int layer = 0; layer < 2; ++layer) {
            if (x2 && proj_x2_ok((int64_t)nb * frames, in_dim, H)) {
                // the layer's input rows as half planes (clip by clip where a clip's rows are not adjacent to the next one's: long form),
                // then gi = x W_ih^T + b_ih as three f16 products
                const float *xf = reinterpret_cast<const float *>(x);
                if (stride_a == (int64_t)frames * lda) {
                    LA_TRY(la_split_f16x2(xf, lda, nb * frames, in_dim, b.planes, in_dim, b.inv, stream));
                } else {
                    for (int c = 0; c < nb; ++c)
                        LA_TRY(la_split_f16x2(xf + (int64_t)c * stride_a, lda, frames, in_dim, static_cast<unsigned char *>(b.planes) + (size_t)c * frames * 2 * in_dim * 2,
                                              in_dim, b.inv + (size_t)c * frames, stream));
                }
                LA_TRY(la_gemm_f16x2(nb * frames, 6 * H, in_dim, 1, b.planes, b.inv, w->w_ih_x2[layer], w->w_ih_x2s[layer], b.gi, 6 * H, w->b_ih[layer],
                                     nullptr, 0, LA_EPI_BIAS, stream));
            } else
            LA_TRY(la_gemm(dt, frames, 6 * H, in_dim, nb, x, lda, stride_a, w->w_ih[layer], b.gi, 6 * H, (int64_t)frames * 6 * H, w->b_ih[layer],
                           nullptr, 0, 0, LA_EPI_BIAS | out_f32, stream));
            LA_TRY(la_gru_layer(dt, b.gi, w->w_hh[layer], w->b_hh[layer], b.gru[layer], layer == 1 ? b.act : nullptr, nb, frames, H, b.gru_ws,
                                b.gru_ws_bytes, timeout_flag, stream));
            x = static_cast<const unsigned char *>(b.gru[layer]);
            lda = 2 * H; stride_a = (int64_t)frames * 2 * H; in_dim = 2 * H;
        }
        // the slice's fused Linear + emission prep and its DP, on the slice's rows of the caller's label / result arrays
        float *em = emissions_out ? emissions_out + (int64_t)b0 * em_clip : b.em;
        const int32_t *lab = labels + (int64_t)b0 * labels_stride;
        if (x2)
            LA_TRY(la_fc_emissions_x2(static_cast<const float *>(b.act), 2 * H, static_cast<const float *>(w->w_fc), w->b_fc, w->w_fc_x2, w->w_fc_x2s, nb, frames,
                                      2 * H, w->vocab, variant, lab, labels_stride, n_labels + b0, max_labels, em, em_clip, max_labels + 1, b.fc_ws,
                                      b.fc_ws_bytes, stream));
        else
        LA_TRY(la_fc_emissions(dt, b.act, 2 * H, w->w_fc, w->b_fc, nb, frames, 2 * H, w->vocab, variant, lab, labels_stride, n_labels + b0, max_labels,
                               em, em_clip, max_labels + 1, b.fc_ws, b.fc_ws_bytes, stream));
        LA_TRY(la_viterbi_batch(em, em_clip, max_labels + 1, lab, labels_stride, n_labels + b0, b.n_frames, nb, frames, max_labels,
                                onset + (int64_t)b0 * out_stride, offset + (int64_t)b0 * out_stride, out_stride, final_score + b0, status + b0,
                                b.vit_ws, b.vit_ws_bytes, stream));
    }
    return LA_OK;
}
