/*
 * lyricalign.h -- C ABI of liblyricalign_hip.so (MI355X / gfx950 only).
 *
 * The reference (navi0105/LyricAlignment) is pure Python and has no FFI: its
 * "operator interface" for the alignment hot path is the set of Python calls
 *
 *   whisper.audio.log_mel_spectrogram(audios)        module/align_model.py:84
 *   whisper_model.embed_audio(mel)                   module/align_model.py:91,101,112,137
 *   align_rnn(embed)   (GRU -> Mish -> Linear)        module/align_model.py:35-38,107,115
 *   perform_viterbi_ctc / perform_viterbi            utils/alignment.py:121-188 / :13-71
 *   run_viterbi_core                                 utils/alignment.py:73-119
 *
 * Every entry point below names the call it replaces.  The Python packages in
 * lyricalignment_amd/ (same names and signatures as the reference's modules)
 * bind these symbols with ctypes; INTEGRATION.md shows the stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++ / torch types cross the boundary;
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer; scratch comes from a caller-provided
 *     workspace sized by the matching *_workspace_bytes() query;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     calls only enqueue work: on the inference path (everything the two
 *     model-level entry points la_encoder_forward / la_align_head_forward
 *     and the alignment DP call) they never synchronise, allocate or free;
 *   - return value: la_status.  Nothing throws across the boundary.
 *     Per-utterance outcomes of the DP are reported in a device `status`
 *     array with the same codes.
 *   - re-entrant.  Library-owned state, all of it mutex-guarded:
 *       (1) the optional kernel timer (la_timer_*), for bench.py only;
 *       (2) one scratch buffer per (device, stream, purpose) for two
 *           TRAINING-path calls whose scratch size depends on a run-time split
 *           decision -- float32 la_gemm / la_gemm_ex when they split K over
 *           extra workgroups (few tiles, long K: the text decoder's 80-row
 *           GEMMs) and la_colsum_f32.  Its first use on a stream calls
 *           hipMalloc (>= 1 MiB); growing it calls hipStreamSynchronize on that
 *           stream, hipFree and hipMalloc once; it lives until the process ends.
 *           LA_GEMM_NO_SPLITK=1 keeps la_gemm off it.  No 16-bit (inference)
 *           call touches it.
 */
#ifndef LYRICALIGN_H
#define LYRICALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum la_status {
    LA_OK = 0,
    LA_EINVAL = 1,       /* bad argument (shape / alignment / null pointer)            */
    LA_EINFEASIBLE = 2,  /* a label state is never visited -> reference: ValueError    */
                         /*   ("k is not in list", utils/alignment.py:183)             */
    LA_EEMPTY = 3,       /* utterance without labels -> reference: IndexError (:152)   */
    LA_EHIP = 4,         /* HIP runtime error, see la_last_error()                     */
    LA_ETIMEOUT = 5,     /* a bounded in-kernel wait gave up (persistent GRU kernel)   */
    LA_EUNSUPPORTED = 6  /* shape outside what the gfx950 kernels are built for        */
} la_status;

typedef enum la_dtype {
    LA_F32 = 0,  /* parity mode: f32 storage, f32-input MFMA (exact fmaf chains)      */
    LA_BF16 = 1, /* throughput mode: bf16 operands, f32 accumulate, f32 residual      */
    LA_F16 = 2   /* the same kernels on IEEE half operands (BASELINE configs[3]: "fp16 MFMA"); 10-bit mantissa, range
                    +-65504 -- Whisper was trained in fp16, so its activations fit                                   */
} la_dtype;

/*
 * Modifier bit on the dtype argument of la_attention / la_attention_ex / la_attention_cached and on la_encoder_weights.dtype
 * (16-bit dtypes): the q operand -- the q rows of wqkv and of bqkv -- carries head_dim^-0.5 * log2(e) instead of
 * head_dim^-0.5 (whisper/model.py MultiHeadAttention.qkv_attention scales q and k by head_dim^-0.25 each), so the scores
 * leave the matrix pipe in the exp2 domain.  With LA_BF16 the attention kernel then also starts its running maximum at 0
 * and subtracts nothing until a query needs it (la_attention.hip, FOLD): one vector instruction less per score.
 */
#define LA_Q_LOG2 0x100


typedef enum la_variant {
    LA_VARIANT_PLAIN = 0, /* perform_viterbi:      log_softmax over all V, silence = col 0          */
    LA_VARIANT_CTC = 1    /* perform_viterbi_ctc:  log_softmax over cols 1..V-2, silence = sigmoid  */
                          /*                       of the last column (naive log, clip -1000)       */
} la_variant;

/* ------------------------------------------------------------------------- */
/* library                                                                    */
/* ------------------------------------------------------------------------- */
int la_version(void);                 /* ABI version, currently 2 (la_encoder_block / la_head_weights carry the f16x2 planes) */
const char *la_last_error(void);      /* thread-local text of the last LA_EHIP / LA_EINVAL     */
int la_device_arch_ok(void);          /* 1 if the current device is gfx950                     */

/* Optional per-kernel HIP-event timer (bench.py roofline leg).  When enabled,
 * launches of the kernel family `name` ("gemm_bf16", "attention", ...) on any
 * stream are bracketed by hipEvents; la_timer_read() synchronises those events
 * and returns the summed milliseconds and the launch count since reset.      */
int la_timer_enable(const char *name);
int la_timer_disable(void);
int la_timer_read(double *total_ms, int64_t *launches);
int la_timer_reset(void);
/* Sampling: bracket every `period`-th launch of the family only (an event record is a barrier packet of its own: ~6.6 us of stream
 * idle time each between back-to-back kernels).  la_timer_read_work also returns the summed work (flops for the GEMM families:
 * 2 M N K batch) of the bracketed launches and the number of launches of the family seen since the reset. */
int la_timer_sample(int32_t period);
int la_timer_read_work(double *total_ms, int64_t *timed_launches, double *timed_work, int64_t *all_launches);

/* Library options: the few choices between SHIPPED kernel forms that a caller (or a test) may want to pin.  Each is resolved ONCE per
 * process -- on first use, from the environment variable named below -- into a plain struct that the launch paths read; no launch
 * calls getenv.  la_set_option overrides a value at run time (between launches; not synchronised with launches in flight on other
 * threads), la_get_option reads it back.  Unknown name -> LA_EINVAL.
 *   name              env                 values
 *   "gemm_tile"       LA_GEMM_TILE        0 = by shape (default) | 128 | 256 (the 256 x 128 three-stage tile) | 512 (force the 256 x 256 kernel)
 *   "gemm_loop"       LA_PP_DBG           0 = hand-placed main loop where K % 128 == 0 (default) | 99 = quadrant ping-pong loop
 *                                         everywhere (the loop of the other K; identical bits)
 *   "gemm_splitk"     LA_GEMM_NO_SPLITK   1 = float32 GEMMs with few tiles cut K over batch slots (default) | 0 = never (no library scratch)
 *   "attn_nw"         LA_ATTN_NW          0 = 256-query workgroups from 1024 positions on (default) | 4 = 128-query | 8 = 256-query
 *   "gru_nw"          LA_GRU_NW           0 = 8-wave workgroups (default) | 4 = 4-wave workgroups of the persistent recurrence
 *   "gru_fence"       LA_GRU_FENCE        0 = write-through hand-off (default) | 1 = release / acquire fences
 *   "gru_handoff"     LA_GRU_HANDOFF      0 = data-tagged 8-byte granules, no counter, from 17 clips on (default; 16-bit, 8-wave workgroups;
 *                                         float32 training sweeps: always) | 1 = the counter form everywhere (write-through stores, drain,
 *                                         barrier, counter add; poll, barrier, loads) | 2 = granules for every batch
 *   "gru_poll_delay"  LA_GRU_POLL_DELAY   0 (default) = poll at once | n = n x 64 clocks of sleep before a step's first poll of the granule hand-off
 *   "viterbi_dpp"     LA_VITERBI_NO_DPP   1 = DPP wave shifts (default) | 0 = the LDS-exchange form
 *   "head_clip_cap"   LA_HEAD_CLIP_CAP    0 = by residency (default) | n = clips per head launch set of la_align_head_forward
 *   "ln_fusion"       LA_LN_FUSION        1 = LayerNorm folded into the 16-bit encoder GEMMs where they run on the 256 x 256 kernel | 0 = never
 *   "resid_split"     LA_RESID_SPLIT      1 = the 16-bit encoder keeps its residual stream split (hi 16-bit + lo byte) | 0 = f32 stream
 *   "gru_timeout_us"  LA_GRU_TIMEOUT_US   0 = 3 s (default) | n = bound of one inter-workgroup wait of the persistent GRU kernels in microseconds (a launch
 *                                         whose workgroups are not all co-resident never completes a hand-off: the bounded wait sets timeout_flag)
 *   "gru_fault_step"  LA_GRU_FAULT_STEP   test hook, ONE launch (the launch that reads it clears it): workgroup (0, 0, 0) of the next GRU forward launch
 *                                         leaves at step n without publishing, as if it had never become resident -- the other workgroups
 *                                         time out.  0 = off (default)
 *   "x2_inference"    LA_X2_INFERENCE     1 = float32 inference (la_encoder_forward / la_align_head_forward with LA_F32 weights that carry f16x2
 *                                         planes) on the f16 matrix pipe at float32 accuracy (default) | 0 = the float32-MFMA kernels (A/B partner)
 * A library built with -DLA_EXPERIMENTS (tools/build_variant.sh; la_has_experiments() == 1) additionally carries the measured-slower
 * kernel structures of rounds 2-4 and their per-launch developer switches; the shipped library has neither. */
int la_set_option(const char *name, int64_t value);
int la_get_option(const char *name, int64_t *value);
int la_has_experiments(void);

/* ------------------------------------------------------------------------- */
/* forced-alignment DP   (replaces utils/alignment.py:73-119 + :141-185)      */
/* ------------------------------------------------------------------------- */
/*
 * Emissions use the COMPACT layout the fused head produces:
 *   em[b][t][0]     = log-prob of silence at frame t      (reference: ls[t][0])
 *   em[b][t][1 + n] = log-prob of label n's class          (reference: lp[t][label[n]-1])
 * (equal neighbouring labels therefore carry identical columns, as they read one class column)
 * float32, strides given in elements.  labels[b][n] are class ids (only equality
 * of neighbours is used, utils/alignment.py:104); n_labels[b] = L_b, n_frames[b] = T_b.
 *
 * Scores are accumulated in float64 with the reference's comparison order and
 * tie rules; backpointers are 2-bit offsets {0,1,2}.  Outputs are INTEGER
 * frame indices: onset[b][n] = first frame in state 2n+1, offset[b][n] = last
 * such frame + 1 (the Python layer multiplies by hop_size_second exactly as
 * utils/alignment.py:185 does).  status[b] is LA_OK / LA_EINFEASIBLE / LA_EEMPTY
 * / LA_EINVAL.  Rows n >= L_b of onset/offset are set to -1.
 * Limits: max_labels <= 4095 (one lane per lattice state up to 1024 states; beyond that 1024 threads sweep 2 / 4 / 8
 * consecutive states each, backpointers in `workspace`); LA_EUNSUPPORTED above -- the reference's numba loop has no limit.
 * `workspace` (8-byte aligned) is needed when la_viterbi_workspace_bytes() > 0: long utterances or > 511 labels.
 */
int la_viterbi_workspace_bytes(int32_t batch, int32_t max_frames, int32_t max_labels, size_t *bytes);

int la_viterbi_batch(const float *em, int64_t em_batch_stride, int64_t em_row_stride,
                     const int32_t *labels, int32_t labels_stride,
                     const int32_t *n_labels, const int32_t *n_frames,
                     int32_t batch, int32_t max_frames, int32_t max_labels,
                     int32_t *onset, int32_t *offset, int32_t out_stride,
                     double *final_score, int32_t *status,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * run_viterbi_core(dp, bt, lp, ls, label) (utils/alignment.py:73-119) for ONE utterance, for callers
 * that want the full matrices: row 0 of dp [T][S] (float64, S = 2L+1) is READ as the caller initialised
 * it (the reference does the same, :144-152); rows >= 1 of dp and of bt [T][S] (int64 predecessor state)
 * are written.  em is the compact emission layout of that utterance ([T][>= L+1]).  n_labels / n_frames
 * are 1-element device arrays holding the same L and T as the host arguments; scratch_i32 has 2L+1
 * elements, scratch_f64 one.
 */
int la_viterbi_core(const float *em, int64_t em_row_stride, const int32_t *labels, int32_t n_labels_host,
                    int32_t n_frames_host, const int32_t *n_labels, const int32_t *n_frames,
                    double *dp, long long *bt, int32_t *scratch_i32, double *scratch_f64,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* emission prep from materialised logits                                     */
/*   (replaces utils/alignment.py:123-134 [CTC] and :14-20 [plain])           */
/* ------------------------------------------------------------------------- */
/*
 * logits [batch][frames][vocab] float32 (device).  Computes the row normaliser
 * over the variant's column range and writes only the columns the DP reads:
 * em[b][t][0] and em[b][t][1+n] for n < n_labels[b] (compact layout above).
 */
int la_emissions_from_logits(const float *logits, int64_t batch_stride, int64_t row_stride,
                             int32_t batch, int32_t frames, int32_t vocab, int32_t variant,
                             const int32_t *labels, int32_t labels_stride, const int32_t *n_labels,
                             int32_t max_labels,
                             float *em, int64_t em_batch_stride, int64_t em_row_stride, void *stream);

/* ------------------------------------------------------------------------- */
/* log-mel front end   (replaces whisper.audio.log_mel_spectrogram + pad_or_trim, */
/*                      call sites module/align_model.py:84,89,100,109)           */
/* ------------------------------------------------------------------------- */
/*
 * audio [batch][n_samples] float32 -> mel [batch][80][n_frames], n_frames =
 * n_samples / 160, float32.  The "-8" floor uses the max over the WHOLE batch
 * tensor, as upstream whisper does.  mel_filters is the [80][201] Slaney table
 * (device), window the 400-tap periodic Hann (device).
 */
int la_logmel_workspace_bytes(int32_t batch, int32_t n_samples, size_t *bytes);
int la_logmel_f32(const float *audio, int32_t batch, int32_t n_samples,
                  const float *mel_filters, const float *window,
                  float *mel, int64_t mel_batch_stride, int64_t mel_row_stride,
                  void *workspace, size_t workspace_bytes, void *stream);
/*
 * The same with the constants (windowed DFT matrix in MFMA fragment order, padded
 * filter bank) built once per (mel_filters, window) pair into a caller-owned
 * device buffer of la_logmel_constants_bytes(): 2 launches per call instead of 3.
 * whisper builds its window and loads its filter asset once per process the same
 * way (whisper.audio.mel_filters is lru-cached).
 */
int la_logmel_constants_bytes(size_t *bytes);
int la_logmel_constants(const float *mel_filters, const float *window,
                        void *consts, size_t consts_bytes, void *stream);
int la_logmel_f32_prepared(const float *audio, int32_t batch, int32_t n_samples,
                           const void *consts,
                           float *mel, int64_t mel_batch_stride, int64_t mel_row_stride,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* building blocks of embed_audio / align_rnn                                 */
/* ------------------------------------------------------------------------- */
/* epilogue flags for la_gemm */
enum {
    LA_EPI_BIAS = 1,      /* + bias[n] (f32)                                          */
    LA_EPI_GELU = 2,      /* GELU after the bias.  f32 results (float32 mode, LA_EPI_OUT_F32): the erf form, |error| <= 1.2e-7.
                           * 16-bit results: x * sigmoid(x (a + b x^2 + c x^4)), a minimax fit to x Phi(x): max ABSOLUTE error 2.5e-5
                           * (below half a bf16 ulp for |gelu| >= 1e-2; the relative error is unbounded in the negative tail, where
                           * |gelu| < 1e-2 -- whisper's F.gelu is the erf form).  LA_EPI_GELU_ERF selects the erf form there too. */
    LA_EPI_RESIDUAL = 4,  /* + residual[m][n] (f32, own strides; batch stride may be 0) */
    LA_EPI_OUT_F32 = 8,   /* C is f32 regardless of the operand dtype                 */
    LA_EPI_MISH = 16,     /* x * tanh(softplus(x)) after the bias                     */
    LA_EPI_RES_GELU_GRAD = 32, /* la_gemm_f16x2 only, with LA_EPI_RESIDUAL: the result is MULTIPLIED by gelu'(residual[m][n]) instead of having
                           * it added -- the input gradient of the MLP's second Linear times the GELU derivative at the saved
                           * pre-activation (la_gelu_bwd_f32 without its pass over the [rows][4 d] buffer) */
    LA_EPI_GELU_ERF = 4096, /* with LA_EPI_GELU and a 16-bit result: the erfc-based form (max absolute error 1.3e-6, 20 x closer to
                           * F.gelu than the sigmoid fit) at ~40 % more epilogue issue slots.  (LA_GELU_PK=1 in the environment sets
                           * it on every launch: the developer A/B switch.) */
    /* operand layout flags of la_gemm_ex (float32 only), OR-ed into the same word: the operand is stored TRANSPOSED,
     * [K][rows] with pitch lda / ldw >= rows (rows % 4 == 0).  With a flag set K may be any length (a K-contiguous operand
     * must then have a pitch >= K rounded up to 32); both set = the weight-gradient shape
     * dW[n][k] = sum_m dY[m][n] X[m][k], which reads dY and X as they are.  Replaces a transpose pass per operand. */
    LA_GEMM_TRANS_A = 512,
    LA_GEMM_TRANS_W = 1024
};

/*
 * C[z][m][n] = epi( sum_k A[z][m][k] * W[n][k] ),  z < batch.
 * A: `dtype`, row stride lda (elements; may be < K: the conv-as-GEMM views
 * overlap rows), batch stride strideA.  W: `dtype`, [N][K] row-major (nn.Linear
 * layout).  C: `dtype` (or f32 with LA_EPI_OUT_F32), row stride ldc, batch
 * stride strideC.  K % 64 == 0 (bf16) / K % 32 == 0 (f32); A, W, C 16-byte aligned;
 * M, N arbitrary (edge tiles are predicated).
 * Replaces every nn.Linear / nn.Conv1d of whisper.model.AudioEncoder and the
 * GRU input projections of module/align_model.py:23-28.
 */
int la_gemm(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch,
            const void *A, int64_t lda, int64_t strideA,
            const void *W,
            void *C, int64_t ldc, int64_t strideC,
            const float *bias,
            const float *residual, int64_t ldr, int64_t strideR,
            int32_t epilogue, void *stream);

/* y[m][:] = LayerNorm(x[m][:]) * gamma + beta, eps 1e-5; x f32 [M][d], y `out_dtype`. */
int la_layernorm(const float *x, int64_t ldx, int32_t M, int32_t d,
                 const float *gamma, const float *beta,
                 void *y, int64_t ldy, int32_t out_dtype, void *stream);

/*
 * Non-causal multi-head self-attention over packed QKV rows.
 * qkv [batch*frames][3*n_head*64] (`dtype`): columns [q | k | v], head h at
 * h*64; q is expected PRE-SCALED by head_dim^-0.5 (folded into the projection
 * weights at pack time; the reference scales q and k by head_dim^-0.25 each,
 * whisper MultiHeadAttention.qkv_attention).  softmax in f32.
 * out [batch*frames][n_head*64] (`dtype`).  head_dim is 64 for every Whisper size.
 */
int la_attention(int32_t dtype, const void *qkv, int64_t ld_qkv, void *out, int64_t ld_out,
                 int32_t batch, int32_t frames, int32_t n_head, void *stream);

/* General form for the Whisper text decoder (whisper.model.TextDecoder; module/align_model.py:118-121): separate
 * q [batch*q_len][>= n_head*64] and k / v [batch*kv_len][>= n_head*64] row sets (k and v share ld_kv), q pre-scaled.
 * causal != 0: key j is visible to query i iff j <= i (decoder self-attention; requires q_len == kv_len). */
int la_attention_ex(int32_t dtype, const void *q, int64_t ld_q, const void *k, const void *v, int64_t ld_kv,
                    void *out, int64_t ld_out, int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head,
                    int32_t causal, void *stream);

/* Attention against a key / value cache (autoregressive decoding: the next row of SURVEY 8f, inference_transcript.py): clip
 * b's keys / values are rows [b*kv_batch_rows, b*kv_batch_rows + kv_len) of k / v (the cache holds kv_batch_rows >= kv_len
 * rows per clip), its queries rows [b*q_batch_rows, +q_len) of q, out rows are packed [b*q_len + i].  causal with q_len == 1
 * (one new token) needs no mask; causal with q_len == kv_len is la_attention_ex's mask. */
int la_attention_cached(int32_t dtype, const void *q, int64_t ld_q, int64_t q_batch_rows, const void *k, const void *v,
                        int64_t ld_kv, int64_t kv_batch_rows, void *out, int64_t ld_out, int32_t batch, int32_t q_len,
                        int32_t kv_len, int32_t n_head, int32_t causal, void *stream);

/* out[r] = argmax_c x[r][c] (first maximum), int64: the greedy token choice over the decoder logits. */
int la_argmax_rows_f32(const float *x, int64_t ld, int32_t rows, int32_t cols, int64_t *out, void *stream);

/* The k (<= 8) largest values of each row, their indices (first maximum on ties) and the row's log-sum-exp:
 * log-probability of candidate j = vals[r][j] - lse[r] (beam search over the decoder logits). */
int la_topk_rows_f32(const float *x, int64_t ld, int32_t rows, int32_t cols, int32_t k, float *vals, int64_t *idx, float *lse,
                     void *stream);

/* x[b*n_tok + i][:] = token_embedding[tokens[b][i]][:] + positional_embedding[i][:]   (f32 tables, f32 out);
 * TextDecoder.forward's first line.  tokens int64 [batch][n_tok]. */
int la_embed_tokens(const int64_t *tokens, int32_t batch, int32_t n_tok, const float *token_embedding, int32_t n_vocab,
                    const float *positional_embedding, int32_t d, float *x, void *stream);

/* mel [batch][n_mels][frames] f32 -> channels-last zero-padded rows for the conv-as-GEMM view:
 * out[b][1 + t][c] (`dtype`), row pitch `c_pad`, (frames + 2) rows per clip, rows 0 and frames+1
 * and channels >= n_mels zeroed. */
int la_mel_to_rows(const float *mel, int64_t mel_batch_stride, int64_t mel_row_stride,
                   int32_t batch, int32_t n_mels, int32_t frames,
                   void *out, int32_t c_pad, int32_t dtype, void *stream);

/*
 * One bidirectional GRU layer recurrence (nn.GRU gate order r,z,n;
 * module/align_model.py:23-28).  gi [batch][frames][2][3H] f32 holds the input
 * projections W_ih x + b_ih of both directions; w_hh [2][3H][H] (`dtype`),
 * b_hh [2][3H] f32.  out [batch][frames][2H] (`dtype`) receives h_t (forward in
 * columns 0..H-1, reverse in H..2H-1); out_mish (optional, same shape) receives
 * Mish(h_t) (module/align_model.py:37).  Persistent kernel: 2 * H/128 (16-bit modes, H % 128 == 0, H <= 384; else 2 * H/64) workgroups
 * per 16 clips exchange h in `workspace` (zeroed on the stream by this call): by default as data-tagged 8-byte granules
 * {h, h | step} polled by the consumers (16-bit modes; option gru_handoff = 1: through `out` with write-through (sc1)
 * stores and per-step arrival counters).  H % 64 == 0;
 * at most 224 workgroups may be co-resident (16-bit modes, H=384: 8-wave workgroups, 2 * 3 per 16 clips).
 * dtype LA_F32, more than 4 clips, option x2_inference = 1 (round 6): the recurrent product h W_hh^T runs as three f16 MFMA products
 * over split operands at float32 accuracy (W_hh rows scaled and split into two half fragment sets resident in registers, h exchanged as
 * {hi | lo, step} granules): 4.4 instead of 11.4 us per step at 16 clips; fewer clips keep the float32 kernel's v_fma path.
 * `timeout_flag` (device int32, optional) is set non-zero if a bounded wait (option gru_timeout_us, default 3 s) gave up: the launch set was
 * not co-resident and `out` is garbage.  Recover by calling again after hipDeviceSynchronize (alone on the device the set is co-resident);
 * INTEGRATION.md "GRU time-out and recovery".
 */
int la_gru_workspace_bytes(int32_t batch, int32_t frames, int32_t hidden, size_t *bytes);
int la_gru_layer(int32_t dtype, const float *gi, const void *w_hh, const float *b_hh,
                 void *out, void *out_mish, int32_t batch, int32_t frames, int32_t hidden,
                 void *workspace, size_t workspace_bytes, int32_t *timeout_flag, void *stream);

/*
 * Fused head tail: Linear(2H -> V) + emission prep, WITHOUT materialising the
 * [batch][frames][V] logits (replaces module/align_model.py:38 followed by
 * utils/alignment.py:123-134 / :14-20 and the device->host copy at
 * inference_alignment.py:161).  act [batch*frames][2H] (`dtype`) = Mish(GRU out);
 * w_fc [V][2H] (`dtype`), b_fc [V] f32.  Writes compact emissions (layout above).
 */
int la_fc_emissions_workspace_bytes(int32_t dtype, int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab,
                                    int32_t max_labels, size_t *bytes);
int la_fc_emissions(int32_t dtype, const void *act, int64_t ld_act, const void *w_fc, const float *b_fc,
                    int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab, int32_t variant,
                    const int32_t *labels, int32_t labels_stride, const int32_t *n_labels,
                    int32_t max_labels,
                    float *em, int64_t em_batch_stride, int64_t em_row_stride,
                    void *workspace, size_t workspace_bytes, void *stream);
/* la_fc_emissions for FLOAT32 rows with the normaliser product (the full [batch*frames] x V Linear) on the f16 matrix pipe at float32
 * accuracy: w_fc_x2 / w_fc_x2s = W_fc as f16x2 planes [V][2][in_dim] (la_split_f16x2, kp = in_dim) and their per-row inverse scales; act is split
 * inside (workspace).  Same results to float32 accuracy; shapes outside the 256 x 256 kernel's domain (in_dim % 128, < 192 tiles) take
 * la_fc_emissions' float32 kernel. */
int la_fc_emissions_x2_workspace_bytes(int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab, int32_t max_labels, size_t *bytes);
int la_fc_emissions_x2(const float *act, int64_t ld_act, const float *w_fc, const float *b_fc, const void *w_fc_x2, const float *w_fc_x2s,
                       int32_t batch, int32_t frames, int32_t in_dim, int32_t vocab, int32_t variant,
                       const int32_t *labels, int32_t labels_stride, const int32_t *n_labels, int32_t max_labels,
                       float *em, int64_t em_batch_stride, int64_t em_row_stride, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* fine-tune path: losses on the align logits and the optimizer step          */
/* ------------------------------------------------------------------------- */
/*
 * compute_ce_loss(compute_sil=True) + compute_ctc_loss (train_multitask.py:587-633), forward and gradient with
 * respect to the logits.  logits [batch][frames][row_stride >= vocab+1] f32: columns 0..vocab-1 are the word classes
 * (column 0 = CTC blank), column `vocab` the silence logit (vocab = 21128 in the reference).
 *   frame_labels [batch*frames] i32: class id (>= 1) of the frame or -100 (already trimmed / padded to `frames`,
 *                train_multitask.py:596-603); the reference's "-1" shift (:607) is folded into the column choice.
 *   ctc_labels [batch][labels_stride] i32 class ids, n_labels[b] of them valid (target_length, :629).
 * losses[0] = word CE (mean over frames with a label), losses[1] = silence BCE (mean over all frames),
 * losses[2] = CTC (mean over the batch of nll_b / n_labels[b]; +inf if an utterance is infeasible).
 * dlogits (optional, same layout) receives scale * d(losses[0]+losses[1]+losses[2]) / dlogits, restricted to the
 * requested terms (use_ce / use_ctc).  The alpha/beta lattice runs one workgroup per utterance, one lane per state.
 */
int la_multitask_loss_workspace_bytes(int32_t batch, int32_t frames, int32_t max_labels, size_t *bytes);
int la_multitask_loss(const float *logits, int64_t batch_stride, int64_t row_stride, int32_t batch, int32_t frames,
                      int32_t vocab, const int32_t *frame_labels, const int32_t *ctc_labels, int32_t labels_stride,
                      const int32_t *n_labels, int32_t max_labels, int32_t use_ce, int32_t use_ctc, float scale,
                      float *losses, float *dlogits, int64_t d_batch_stride, int64_t d_row_stride,
                      void *workspace, size_t workspace_bytes, void *stream);

/*
 * clip_grad_norm_(params, max_norm) + AdamW on flat f32 buffers (train_multitask.py:337-340,683-686).
 * la_grad_sqnorm_f32 ADDS sum(grad^2) into *sum_sq (device double; zero it once, call it per bucket, all-reduce is the
 * caller's: gradients are already averaged over ranks before this).  la_adamw_step_f32 applies
 *   g' = g * grad_prescale * min(1, max_norm / (sqrt(*clip_sum_sq) * grad_prescale + 1e-6))    (clip_sum_sq may be NULL)
 *   p *= 1 - lr*wd;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 */
int la_grad_sqnorm_f32(const float *grad, int64_t n, double *sum_sq, void *stream);
int la_adamw_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int32_t step,
                      const double *clip_sum_sq, float max_norm, float grad_prescale, void *stream);

/*
 * Backward pass of the head (GRU x2 bidirectional -> Mish -> Linear, module/align_model.py:11-40), float32.
 * la_gru_layer_train_fwd = la_gru_layer(LA_F32) that also stores, per step, r, z, n and (W_hn h + b_hn) into
 * gates [batch][frames][2][4H].  la_gru_layer_bwd sweeps the recurrence backwards: dout = gradient w.r.t. the layer
 * output [batch][frames][2H]; writes dgi / dgh [batch][frames][2][3H], the gradients w.r.t. the input / recurrent gate
 * pre-activations, from which the weight gradients are plain GEMMs (dW_ih = dgi^T x, dW_hh = dgh^T h_prev, dx = dgi W_ih).
 * The helpers below express those GEMMs for la_gemm (both operands K-contiguous): zero-padded transposes, column sums
 * (bias gradients), Mish forward / backward, masked scaling (the inter-layer dropout of nn.GRU in train mode).
 */
int la_gru_layer_train_fwd(const float *gi, const float *w_hh, const float *b_hh, float *out, float *gates,
                           int32_t batch, int32_t frames, int32_t hidden, void *workspace, size_t workspace_bytes,
                           int32_t *timeout_flag, void *stream);
int la_gru_layer_bwd(const float *gates, const float *out, const float *dout, const float *w_hh, float *dgi, float *dgh,
                     int32_t batch, int32_t frames, int32_t hidden, void *workspace, size_t workspace_bytes,
                     int32_t *timeout_flag, void *stream);
int la_transpose_pad_f32(const float *in, int64_t ld_in, int32_t rows, int32_t cols, float *out, int64_t ld_out,
                         int32_t out_rows, int32_t out_cols, void *stream);
int la_colsum_f32(const float *in, int64_t ld, int32_t rows, int32_t cols, float *out, void *stream);
int la_mish_f32(const float *x, float *y, int64_t n, void *stream);
int la_mish_bwd_f32(const float *x, const float *dy, float *dx, int64_t n, void *stream);
int la_mask_scale_f32(const float *x, const unsigned char *mask, float scale, float *y, int64_t n, void *stream);

/*
 * LayerNorm folded into the GEMMs on either side of it -- whisper's ResidualAttentionBlock computes
 * x = x + attn(attn_ln(x)); x = x + mlp(mlp_ln(x)) (third-party whisper/model.py; call module/align_model.py:91,101,112,137).
 * In the 16-bit modes, for shapes that run on the 256x256 kernel:
 *   producer  (la_gemm_fused_ln with C2 != NULL, LA_EPI_OUT_F32): the GEMM that writes the f32 residual stream also stores
 *             the same rows rounded to `dtype` into C2 [M][ldc2] -- the RAW operand of the next GEMM;
 *   la_row_stats16 (or ln_part + la_ln_stats_finalize): stats[m] = (mean, 1/sqrt(var + eps)) of those raw rows;
 *   consumer  (ln_stats != NULL): A = raw rows, W = gamma-folded weights W'[n][k] = gamma[k] W[n][k], ln_csum[n] = sum_k W'[n][k],
 *             bias[n] = b[n] + sum_k beta[k] W[n][k]:   LN(x) W^T + b = rstd (x W'^T - mean c) + bias.
 * Replaces the separate LayerNorm pass (read f32, write 16-bit) between them.  LA_EUNSUPPORTED for f32 and for shapes
 * the 256x256 kernel does not take (callers fall back to la_layernorm + la_gemm).
 */
int la_gemm_fused_ln(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda, int64_t strideA,
                     const void *W, void *C, int64_t ldc, int64_t strideC, const float *bias, const float *residual, int64_t ldr,
                     int64_t strideR, int32_t epilogue, void *C2, int64_t ldc2, int64_t strideC2, const float *ln_stats,
                     const float *ln_csum, float *ln_part, void *stream);
/*
 * The residual stream of the 16-bit encoder kept SPLIT (the default of la_encoder_forward wherever the LayerNorm fold applies):
 *   hi [M][ld] `dtype` = x rounded to the operand type -- at the same time the raw A operand of the next folded GEMM,
 *   lo [M][ld] uint8   = the remainder x - hi in steps of ulp(hi) / 254, offset by 128
 * (x to 8 bits below the operand type's last place; 3 + 3 bytes per element through HBM per read-modify-write instead of the
 * 4 + 4 + 2 of an f32 stream with a 16-bit copy).  la_gemm_split: (hi, lo) <- epi(A W^T) + residual, epilogue = LA_EPI_BIAS |
 * LA_EPI_GELU | LA_EPI_RESIDUAL; with LA_EPI_RESIDUAL the residual is `residual` (f32 rows, as la_gemm takes them: the stem's
 * positional embedding) or, when that pointer is NULL, the stream itself, updated in place (x += out-proj, x += mlp:
 * whisper/model.py ResidualAttentionBlock.forward).  ln_part as la_gemm_fused_ln.  LA_EUNSUPPORTED for f32 and for shapes the
 * 256x256 kernel does not take.  la_layernorm_split: la_layernorm over rows of such a stream (ln_post).
 */
int la_gemm_split(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda, int64_t strideA,
                  const void *W, void *hi, void *lo, int64_t ld, int64_t stride, const float *bias, const float *residual,
                  int64_t ldr, int64_t strideR, int32_t epilogue, float *ln_part, void *stream);
int la_layernorm_split(int32_t dtype, const void *hi, const void *lo, int64_t ldx, int32_t M, int32_t d, const float *gamma,
                       const float *beta, void *y, int64_t ldy, int32_t out_dtype, void *stream);
/* producer with ln_part != NULL (N % 64 == 0): also stores, per row and 64-column segment of the 16-bit copy, (mean, sum of
 * squared deviations) into ln_part [N/64][M][2]; la_ln_stats_finalize combines them into stats [M][2] = (mean, rstd) --
 * the same numbers la_row_stats16 computes, without reading the copy back. */
int la_ln_stats_finalize(const float *part, int32_t slots, int32_t M, float eps, float *stats, void *stream);
int la_row_stats16(int32_t dtype, const void *x, int64_t ldx, int32_t M, int32_t d, float eps, float *stats, void *stream);

/* float32 Linear products on the f16 matrix pipe at float32 accuracy ("f16x2"; the fine-tune step's nn.Linear forward and both
 * backward products, train_multitask.py:325-326).  gfx950 multiplies float32 operands at 1/16 of its 16-bit MFMA rate.  A float32
 * matrix, its rows scaled by powers of two (largest magnitude of a row in [2^13, 2^14)), splits exactly into two IEEE-half PLANES
 * (hi = f16(x s), lo = f16(x s - hi): 22 bits), and  A W^T = (sa sw^T) o (A_lo W_hi^T + A_hi W_lo^T + A_hi W_hi^T) + O(2^-22)
 * -- three f16 products accumulated in f32 inside ONE pass of the 256 x 256 kernel over segmented K, the scales applied in its
 * epilogue.  Measured (profiles/r5_kbench_f32emu.txt): 2.5-3.2 x faster than float32 la_gemm on the fine-tune shapes, error against
 * a float64 product SMALLER than float32 la_gemm's own.
 *   la_split_f16x2:    x [rows][cols] f32 (pitch ldx) -> planes [rows][2][kp] f16 (kp >= cols, % 8 == 0; tail columns zero) and
 *                      inv_scale [rows] f32 (1 / s, a power of two).
 *   la_split_f16x2_t:  the same of x^T -- planes_t [cols][2][mp] (mp >= rows, % 16 == 0), inv_scale_t [cols]: the operands of a
 *                      weight gradient dW = dY^T X, whose contraction runs over the rows of dY and X.  Uses library scratch (4 B per column).
 *   la_gemm_f16x2:     C [M][N] f32 (pitch ldc) = epi((A W^T) o sa sw^T); A = planes [M][2][K], W = planes [N][2][K] (K = the padded
 *                      plane length both were split to), epilogue = LA_EPI_BIAS | LA_EPI_GELU (erf) | LA_EPI_RESIDUAL (f32 rows, pitch
 *                      ldr).  slots > 1 cuts K into `slots` equal chunks run as batch slots into library scratch and summed in slot
 *                      order (few tiles, long K: the weight gradients).  Domain: K / slots a multiple of 128 and >= 256, N > 128,
 *                      ceil(M/256) ceil(N/256) slots >= 192 -- else LA_EUNSUPPORTED (the caller keeps float32 la_gemm for small shapes). */
int la_split_f16x2(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale, void *stream);
int la_split_f16x2_t(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t, void *stream);
/* the same with an activation applied to the values before the split (act: 0 = none, 1 = exact-erf GELU): the MLP's hidden operand
 * gelu(u) goes from u into its planes without a float32 copy of its own (forward product and weight gradient) */
int la_split_f16x2_act(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale, int32_t act, void *stream);
int la_split_f16x2_t_act(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t, int32_t act, void *stream);
/* la_split_f16x2_t_act that also returns colsum[c] = sum_r x[r][c] (NULL: not wanted; act must be 0): the bias gradient sum_r dy next to
 * the weight gradient dy^T x whose operand the split makes, from the same pass over dy (float64 partials per row block, added in order:
 * deterministic, as la_colsum_f32) */
int la_split_f16x2_t_colsum(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t, int32_t act,
                            float *colsum, void *stream);
/* An operand that is split both ways (dy: plain for dx = dy w, transposed for dw = dy^T x; an activation: plain in the forward, transposed
 * for its weight gradient) needs no second pass for the transposed planes' scales: la_split_f16x2_max is la_split_f16x2_act that also
 * leaves the operand's largest magnitude in tmax (2048 words ZEROED by the caller: 64 are used, one per 128-byte line, filled with atomicMax), and la_split_f16x2_t_tmax with
 * that tmax gives the transposed planes ONE power-of-two scale for the whole operand (tmax NULL = la_split_f16x2_t_colsum: a scale per
 * column).  Entries more than 2^17 below the operand's maximum lose relative, never absolute (2^-39 of the maximum), precision. */
int la_split_f16x2_max(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes, int64_t kp, float *inv_scale, int32_t act,
                       uint32_t *tmax, void *stream);
int la_split_f16x2_t_tmax(const float *x, int64_t ldx, int32_t rows, int32_t cols, void *planes_t, int64_t mp, float *inv_scale_t, int32_t act,
                          float *colsum, const uint32_t *tmax, void *stream);
int la_gemm_f16x2(int32_t M, int32_t N, int32_t K, int32_t slots, const void *A, const float *sa, const void *W, const float *sw,
                  float *C, int64_t ldc, const float *bias, const float *residual, int64_t ldr, int32_t epilogue, void *stream);
/* la_layernorm (eps 1e-5, statistics in float32 over the row) whose result leaves as the f16x2 planes of the next Linear's operand instead of
 * as float32 rows: planes [rows][2][kp] + inv_scale [rows] as la_split_f16x2 would make them from LN(x) gamma + beta (float32 inference on the
 * f16x2 products: module/align_model.py:72-123 is float32 throughout).  d % 4 == 0, d <= 4096. */
int la_layernorm_f16x2(const float *x, int64_t ldx, int32_t rows, int32_t d, const float *gamma, const float *beta, void *planes, int64_t kp,
                       float *inv_scale, void *stream);

/* Backward-pass building blocks of the Whisper encoder (float32): la_gemm with a row pitch for W and per-batch strides
 * (attention gradients batch over heads inside the packed [T][3d] projections), batched zero-padded transposes, exact-erf
 * GELU forward / backward, LayerNorm backward (dx and dy*xhat, whose column sum is dgamma), row softmax forward /
 * backward on materialised score tiles, the overlap-add of the k=3 convolution gradients, elementwise add. */
int la_gemm_ex(int32_t dtype, int32_t M, int32_t N, int32_t K, int32_t batch, const void *A, int64_t lda, int64_t strideA,
               const void *W, int64_t ldw, int64_t strideW, void *C, int64_t ldc, int64_t strideC, const float *bias,
               int32_t epilogue, void *stream);
int la_transpose_pad_batched_f32(const float *in, int64_t ld_in, int64_t batch_stride_in, int32_t rows, int32_t cols,
                                 float *out, int64_t ld_out, int64_t batch_stride_out, int32_t out_rows, int32_t out_cols,
                                 int32_t batch, void *stream);
int la_gelu_f32(const float *x, float *y, int64_t n, void *stream);
int la_gelu_bwd_f32(const float *x, const float *dy, float *dx, int64_t n, void *stream);
int la_add_f32(const float *a, const float *b, float *y, int64_t n, void *stream);
int la_scale_f32(const float *x, float alpha, float *y, int64_t n, void *stream);
int la_layernorm_bwd_f32(const float *x, const float *dy, const float *gamma, int32_t M, int32_t d, float *dx, float *dy_xhat,
                         void *stream);
/* la_layernorm_bwd_f32 together with the parameter gradients dgamma [d] = sum_r dy xhat and dbeta [d] = sum_r dy: one pass over x and dy for
 * d = 256 ... 2048 in steps of 256 (rows in registers, float64 partials per 128-row block added in order); other widths run
 * la_layernorm_bwd_f32 into `scratch` (M x d floats, required then, may be NULL otherwise) and two la_colsum_f32.  residual [M][d] (or
 * NULL): added to dx -- the gradient that reaches the block's input past it (pre-norm residual connection). */
int la_layernorm_bwd_sums_f32(const float *x, const float *dy, const float *gamma, const float *residual, int32_t M, int32_t d, float *dx,
                              float *dgamma, float *dbeta, float *scratch, void *stream);
/* causal_q_len > 0: row r is query (r mod causal_q_len) and sees keys 0 .. (r mod causal_q_len) + cols - causal_q_len */
int la_softmax_rows_f32(float *s, int64_t ld, int64_t rows, int32_t cols, int32_t causal_q_len, void *stream);
int la_softmax_bwd_rows_f32(const float *p, float *dp, int64_t ld, int64_t rows, int32_t cols, void *stream);
/* Fused float32 attention backward, head_dim 64 (the gradient of whisper/model.py MultiHeadAttention.qkv_attention, reached from
 * train_multitask.py:325-326): q [batch*q_len][ld_q] (pre-scaled by head_dim^-0.5, as la_attention takes it), k / v
 * [batch*kv_len][ld_kv], o = the forward output and dout its gradient [batch*q_len][ld_o / ld_do]; head h owns columns
 * 64 h .. 64 h + 63 of every operand.  Writes dq [batch*q_len][ld_dq] (w.r.t. the pre-scaled q) and dk / dv [batch*kv_len][ld_dkv].
 * Nothing of size q_len x kv_len is materialised: scores are recomputed per 64 x 64 tile (three launches: row statistics,
 * a key-block sweep for dk / dv, a query-block sweep for dq).  workspace: la_attention_bwd_workspace_bytes (2 floats per
 * query row and head).  lse: the forward's row statistic (la_attention_lse_f32) or NULL.  causal != 0: key j is visible to queries i >= j (q_len == kv_len). */
int la_attention_bwd_workspace_bytes(int32_t batch, int32_t q_len, int32_t n_head, size_t *bytes);
int la_attention_bwd_f32(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, const float *o, int64_t ld_o,
                         const float *dout, int64_t ld_do, float *dq, int64_t ld_dq, float *dk, float *dv, int64_t ld_dkv,
                         int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, const float *lse,
                         void *workspace, size_t workspace_bytes, void *stream);
/* The statistics launch of la_attention_bwd_f32 alone (lse unless handed in, D = sum dO o O per query row). */
int la_attention_bwd_stats_f32(const float *q, int64_t ld_q, const float *k, int64_t ld_kv, const float *o, int64_t ld_o, const float *dout,
                               int64_t ld_do, int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal,
                               const float *lse_in, float *lse, float *dvec, void *stream);
/* The float32 training forward: la_attention_ex(LA_F32, ...) that also writes lse [batch][n_head][q_len] = log sum_j exp(s_ij);
 * handed to la_attention_bwd_f32 (`lse`; NULL there = recomputed by one more sweep over the scores). */
int la_attention_lse_f32(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, float *out, int64_t ld_out,
                         int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, float *lse, void *stream);
/* la_attention_lse_f32 with both products on the f16 matrix pipe at float32 accuracy (the f16x2 scheme of la_gemm_f16x2: operands split
 * into half planes with one power-of-two scale per (clip, head), three 32x32x16 f16 MFMAs per product, float32 accumulate and softmax;
 * csrc/la_attention_f16x2.hip).  Same arguments and results; workspace (256-byte aligned) of la_attention_f16x2_workspace_bytes bytes
 * holds the planes of q, k, v. */
int la_attention_f16x2_workspace_bytes(int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, size_t *bytes);
int la_attention_lse_f16x2(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, float *out, int64_t ld_out,
                           int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, float *lse, void *workspace,
                           size_t workspace_bytes, void *stream);
/* la_attention_bwd_f32 with its seven products on the f16 matrix pipe at float32 accuracy (key sweep and query sweep in the
 * register-resident form of la_attention_lse_f16x2; q, k, v, dout split into half planes per clip and head; P split with 2^13, dS with a
 * scale from the bound 2 |dO_i| |v_j|).  Same arguments and results; lse may be NULL (recomputed); workspace (256-byte aligned) of
 * la_attention_bwd_f16x2_workspace_bytes bytes. */
int la_attention_bwd_f16x2_workspace_bytes(int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, size_t *bytes);
int la_attention_bwd_f16x2(const float *q, int64_t ld_q, const float *k, const float *v, int64_t ld_kv, const float *o, int64_t ld_o,
                           const float *dout, int64_t ld_do, float *dq, int64_t ld_dq, float *dk, float *dv, int64_t ld_dkv,
                           int32_t batch, int32_t q_len, int32_t kv_len, int32_t n_head, int32_t causal, const float *lse,
                           void *workspace, size_t workspace_bytes, void *stream);
/* Text-decoder training pieces (whisper/model.py TextDecoder; train_multitask.py:285,308 decoder cross-entropy):
 * gradient of token + learned positional embedding (dtok accumulates, dpos [n][d] is written), and
 * F.cross_entropy(logits [rows][vocab], target, ignore_index=-100, 'mean'): loss2[0] = loss, loss2[1] = 1/count;
 * row_ws holds 2*rows floats; dlogits (optional) = scale * dloss/dlogits. */
int la_embed_tokens_bwd_f32(const float *dx, const int64_t *tokens, int32_t batch, int32_t n, int32_t d, int32_t n_vocab, float *dtok,
                            float *dpos, void *stream);      /* ids clamped to [0, n_vocab) like la_embed_tokens */
int la_cross_entropy_f32(const float *logits, int64_t ld, int32_t rows, int32_t vocab, const int64_t *target, float scale,
                         float *loss2, float *row_ws, float *dlogits, int64_t ld_d, void *stream);
int la_col2im3_f32(const float *dcols, int32_t batch, int32_t t_out, int32_t stride, int32_t channels, float *out,
                   int32_t rows_out, void *stream);

/*
 * Polyphase FIR resampling (audio front end, utils/audio.py:3-20: librosa.load(file, sr=16000)):
 * y[n] = sum_k h[(n + skip)*down - k*up] * x[k], n < n_out.  h is the centred low-pass (host-built Kaiser-windowed sinc,
 * gain `up`, zero pre-padded so that `skip` whole output samples are dropped), as in scipy.signal.resample_poly.
 */
int la_resample_poly_f32(const float *x, int64_t n_in, const float *h, int64_t h_len, int32_t up, int32_t down,
                         int64_t skip, float *y, int64_t n_out, void *stream);

/* elementwise helpers used by the host-side plumbing */
int la_cast_f32_to_bf16(const float *x, void *y, int64_t n, void *stream);
int la_cast_bf16_to_f32(const void *x, float *y, int64_t n, void *stream);

/* ------------------------------------------------------------------------- */
/* model-level entry points: one call per stage of the hot path                */
/* ------------------------------------------------------------------------- */
/*
 * Packed weights (device pointers; layouts = lyricalignment_amd/engine.py pack_encoder / pack_head, which build these
 * structs from an openai-whisper AudioEncoder state_dict and the reference's RNN state_dict):
 *   matrices keep nn.Linear's [out][in] layout in the compute dtype; vectors are f32;
 *   wqkv [3d][d] = rows of query (pre-scaled by head_dim^-0.5; by head_dim^-0.5 log2(e) when la_encoder_weights.dtype carries
 *   LA_Q_LOG2), key, value; bqkv its bias (key part zero, query part scaled like the rows);
 *   conv1_w [d][3][128] (tap-major, mel channels zero-padded 80 -> 128), conv2_w [d][3][d];
 *   *_ln (16-bit modes, optional): the LayerNorm-folded forms la_gemm_fused_ln consumes -- W' = gamma o W rounded to the
 *   compute dtype, c[n] = sum_k W'[n][k], b' = b + W beta; NULL = always the separate LayerNorm pass.
 */
typedef struct la_encoder_block {
    const float *ln1_g, *ln1_b;
    const void *wqkv; const float *bqkv;
    const void *wo; const float *bo;
    const float *ln2_g, *ln2_b;
    const void *w1; const float *b1;
    const void *w2; const float *b2;
    const void *wqkv_ln; const float *cqkv, *bqkv_ln;
    const void *w1_ln; const float *c1, *b1_ln;
    /* float32 mode (ABI 2): the four weight matrices also as f16x2 planes [N][2][K] (la_split_f16x2 of wqkv, wo, w1, w2 with kp = K) and
     * their per-row inverse scales [N] -- all eight given: the block's Linear layers, LayerNorms and attention run on the f16 matrix pipe
     * at float32 accuracy (la_gemm_f16x2, la_layernorm_f16x2, la_attention_lse_f16x2) where the batch brings every product into that
     * kernel's domain; NULL: the float32-MFMA kernels.  Ignored in the 16-bit modes. */
    const void *wqkv_x2; const float *wqkv_x2s;
    const void *wo_x2; const float *wo_x2s;
    const void *w1_x2; const float *w1_x2s;
    const void *w2_x2; const float *w2_x2s;
} la_encoder_block;

typedef struct la_encoder_weights {
    int32_t dtype, d, n_head, n_layer, n_mels;
    const void *conv1_w; const float *conv1_b;
    const void *conv2_w; const float *conv2_b;
    const float *pos;                     /* [1500][d] sinusoids                                   */
    const float *lnp_g, *lnp_b;           /* ln_post                                               */
    const la_encoder_block *blocks;       /* HOST array of n_layer entries                         */
} la_encoder_weights;

/*
 * whisper_model.embed_audio(mel) (module/align_model.py:91,101,112,137 = whisper AudioEncoder.forward): mel [batch][n_mels][3000]
 * f32 -> out [batch*1500][d] rows (`out_dtype`, row pitch ld_out) = ln_post(blocks(conv stem(mel) + positional embedding)).
 * Enqueues the whole kernel sequence on `stream` out of `workspace` (256-byte aligned, la_encoder_workspace_bytes; about
 * 58 MB per clip for Whisper-medium in bf16).  No allocation, no synchronisation.
 */
int la_encoder_workspace_bytes(const la_encoder_weights *w, int32_t batch, size_t *bytes);
int la_encoder_forward(const la_encoder_weights *w, const float *mel, int64_t mel_batch_stride, int64_t mel_row_stride,
                       int32_t batch, void *out, int64_t ld_out, int32_t out_dtype, void *workspace, size_t workspace_bytes,
                       void *stream);

typedef struct la_head_weights {
    int32_t dtype, hidden, in_dim, vocab, n_layers;   /* n_layers = 2, bidirectional (module/align_model.py:23-28) */
    const void *w_ih[2]; const float *b_ih[2];        /* per layer [6H][in] (forward rows, then reverse), [6H]     */
    const void *w_hh[2]; const float *b_hh[2];        /* per layer [2][3H][H], [2][3H]                              */
    const void *w_fc; const float *b_fc;              /* [V][2H], [V]                                               */
    /* float32 mode (ABI 2): the input projections and the output Linear also as f16x2 planes ([6H][2][in], [V][2][2H]; la_split_f16x2 with
     * kp = the matrix's own K) + per-row inverse scales -- given: those products and the recurrence's W_hh h run on the f16 matrix pipe at
     * float32 accuracy; NULL: the float32-MFMA kernels.  Ignored in the 16-bit modes. */
    const void *w_ih_x2[2]; const float *w_ih_x2s[2];
    const void *w_fc_x2; const float *w_fc_x2s;
} la_head_weights;

/*
 * align_rnn(embed) + perform_viterbi(_ctc) (module/align_model.py:35-38; utils/alignment.py:13-71,121-188) without the
 * logits: feats rows [.][ld_feats] (compute dtype), clip b at rows b*clip_stride_rows .. +frames -> 2 x {input-projection
 * GEMM, persistent BiGRU recurrence} -> Mish -> fused Linear + emission prep -> batched DP -> onset / offset frames
 * (same outputs as la_viterbi_batch).  emissions_out (optional) [batch][frames][max_labels+1] f32 receives the compact
 * emissions.  More clips than one launch set of the recurrence takes (256; 144 in float32 at hidden 384) run as
 * consecutive slices inside this call.  `timeout_flag`: see la_gru_layer.
 */
int la_align_head_workspace_bytes(const la_head_weights *w, int32_t batch, int32_t frames, int32_t max_labels, size_t *bytes);
int la_align_head_forward(const la_head_weights *w, const void *feats, int64_t ld_feats, int64_t clip_stride_rows,
                          int32_t batch, int32_t frames, int32_t variant, const int32_t *labels, int32_t labels_stride,
                          const int32_t *n_labels, int32_t max_labels, int32_t *onset, int32_t *offset, int32_t out_stride,
                          double *final_score, int32_t *status, float *emissions_out, void *workspace, size_t workspace_bytes,
                          int32_t *timeout_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LYRICALIGN_H */
