"""whisper's `transcribe` / `DecodingTask` on the HIP decoder (inference_transcript.py:72-104 calls
`model.transcribe(audio, task='transcribe', language='zh', beam_size=5)`).

What is here (restated from the published algorithm of whisper/decoding.py and whisper/transcribe.py; openai-whisper is not in this
image, so this is NOT pinned to it -- tests/test_gpu_transcribe.py holds the rules to hand-worked cases and to their invariants):
  * the logit filters of DecodingTask -- SuppressBlank, SuppressTokens, ApplyTimestampRules -- applied to the device logits of every
    step (masking is data movement; the row reductions the timestamp rule needs come from la_topk_rows_f32);
  * greedy / temperature sampling (best_of) and beam search (the beam bookkeeping of AlignEngine.decode_beam) with those filters,
    the no-speech probability from the logits at the start-of-transcript position, the maximum-likelihood ranker;
  * transcribe(): the sliding 30 s window, temperature fallback (compression-ratio / average-log-probability thresholds),
    no-speech skipping, timestamp-token segments and seeking, conditioning on the previous window's text.
What is NOT here: the vocabulary DATA.  whisper's tiktoken rank files are assets of openai-whisper and not in this image;
lyricalignment_amd/tokenizer.py (the whisper.tokenizer stand-in: byte-pair codec + whisper's Tokenizer surface) loads them from
$LA_WHISPER_ASSETS / an installed openai-whisper / an explicit path, and decode / transcribe pick that up by themselves
(resolve_tokenizer); an openai-whisper Tokenizer object may be passed instead.  Without the file, `text` is None, the compression
ratio is taken over the token ids' byte stream, and `suppress_tokens="-1"` (whisper's non-speech symbol list, which is derived
from the vocabulary's text) only covers the special tokens.  The special-token ids below
are constants of whisper's two vocabularies."""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .whisper_compat import HOP_LENGTH, N_FRAMES, SAMPLE_RATE, pad_or_trim

LANGUAGES = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi ml cy sk te "
             "fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps "
             "tk nn mt sa lb my bo tl mg as tt haw ln ha ba jw su").split()
# language names whisper also accepts for `language=` (whisper/tokenizer.py LANGUAGES values and TO_LANGUAGE_CODE aliases), in the
# order of the codes above
LANGUAGE_NAMES = dict(zip(
    [n.replace("_", " ") for n in
     ("english chinese german spanish russian korean french japanese portuguese turkish polish catalan dutch arabic swedish italian "
      "indonesian hindi finnish vietnamese hebrew ukrainian greek malay czech romanian danish hungarian tamil norwegian thai urdu "
      "croatian bulgarian lithuanian latin maori malayalam welsh slovak telugu persian latvian bengali serbian azerbaijani slovenian "
      "kannada estonian macedonian breton basque icelandic armenian nepali mongolian bosnian kazakh albanian swahili galician marathi "
      "punjabi sinhala khmer shona yoruba somali afrikaans occitan georgian belarusian tajik sindhi gujarati amharic yiddish lao uzbek "
      "faroese haitian_creole pashto turkmen nynorsk maltese sanskrit luxembourgish myanmar tibetan tagalog malagasy assamese tatar "
      "hawaiian lingala hausa bashkir javanese sundanese").split()], LANGUAGES))
LANGUAGE_NAMES.update({"burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb", "pushto": "ps",
                       "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si", "castilian": "es", "mandarin": "zh"})
TIME_PRECISION = 0.02          # seconds per timestamp token = 2 mel frames = one encoder frame
INPUT_STRIDE = 2               # mel frames per encoder frame (N_FRAMES // n_audio_ctx)


@dataclass
class TokenizerSpec:
    """Special-token ids of whisper's vocabularies (multilingual: 51865 entries; English-only: 51864).  `codec`: optional object
    with .decode(list[int]) -> str (and .encode(str) -> list[int] for text prompts); `non_speech_ids`: whisper's
    tokenizer.non_speech_tokens (needs the vocabulary's text) if the caller has them."""
    multilingual: bool = True
    codec: Optional[object] = None
    non_speech_ids: Tuple[int, ...] = ()
    blank_id: int = 220                      # encode(" ") in the GPT-2 byte-pair vocabulary both are built on

    @property
    def eot(self) -> int: return 50257 if self.multilingual else 50256
    @property
    def sot(self) -> int: return self.eot + 1
    @property
    def n_lang(self) -> int: return len(LANGUAGES)       # the language tokens exist in both vocabularies (51864 = 50257 + 99 + 6 + 1501 + 1)
    @property
    def translate(self) -> int: return self.sot + 1 + self.n_lang
    @property
    def transcribe(self) -> int: return self.translate + 1
    @property
    def sot_lm(self) -> int: return self.translate + 2
    @property
    def sot_prev(self) -> int: return self.translate + 3
    @property
    def no_speech(self) -> int: return self.translate + 4
    @property
    def no_timestamps(self) -> int: return self.translate + 5
    @property
    def timestamp_begin(self) -> int: return self.translate + 6

    def language_token(self, language: str) -> int:
        if not self.multilingual:
            raise ValueError("the English-only vocabulary has no language tokens")
        return self.sot + 1 + LANGUAGES.index(language)

    def sot_sequence(self, language: Optional[str], task: str) -> Tuple[int, ...]:
        seq = [self.sot]
        if self.multilingual:
            seq += [self.language_token(language or "en"), self.transcribe if task == "transcribe" else self.translate]
        return tuple(seq)

    def decode(self, tokens: Sequence[int]) -> Optional[str]:
        if self.codec is None:
            return None
        return self.codec.decode([int(t) for t in tokens if int(t) < self.eot])


@dataclass
class DecodingOptions:
    """whisper.DecodingOptions (same names and defaults)."""
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    prompt: Optional[Sequence[int]] = None          # token ids of the previous context (after <|startofprev|>)
    prefix: Optional[Sequence[int]] = None          # token ids to prefix the current context with
    suppress_tokens: Optional[Sequence[int] | str] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0


@dataclass
class DecodingResult:
    tokens: List[int] = field(default_factory=list)
    text: Optional[str] = None
    avg_logprob: float = float("nan")
    no_speech_prob: float = float("nan")
    temperature: float = float("nan")
    compression_ratio: float = float("nan")
    language: Optional[str] = None


def compression_ratio(text_or_tokens) -> float:
    """len(utf-8 bytes) / len(zlib(bytes)) of the text (whisper/utils.py); of the token ids' little-endian bytes without a vocabulary."""
    if isinstance(text_or_tokens, str):
        b = text_or_tokens.encode("utf-8")
    else:
        b = np.asarray(list(text_or_tokens), dtype="<u2" if max(list(text_or_tokens) + [0]) < 65536 else "<u4").tobytes()
    return len(b) / len(zlib.compress(b)) if b else 1.0


# ------------------------------------------------------------------------------------------------------------------ logit filters
NEG_INF = float("-inf")


def suppress_blank_(logits: torch.Tensor, n_sampled: int, tok: TokenizerSpec) -> None:
    """SuppressBlank: at the first sampled position neither " " nor <|endoftext|> may be chosen."""
    if n_sampled == 0:
        logits[:, [tok.blank_id, tok.eot]] = NEG_INF


def suppressed_token_ids(options: DecodingOptions, tok: TokenizerSpec) -> List[int]:
    """DecodingTask._get_suppress_tokens: "-1" expands to the tokenizer's non-speech symbols; the special tokens that must
    never be sampled are always added."""
    st = options.suppress_tokens
    if isinstance(st, str):
        st = [int(t) for t in st.split(",") if t.strip()]
    ids = list(st or [])
    if -1 in ids:
        ids = [t for t in ids if t >= 0] + list(tok.non_speech_ids)
    ids += [tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech]
    return sorted(set(int(t) for t in ids))


def apply_timestamp_rules_(logits: torch.Tensor, sampled: Sequence[Sequence[int]], tok: TokenizerSpec,
                           max_initial_timestamp_index: Optional[int]) -> None:
    """ApplyTimestampRules.  logits [N, V] f32 on the device (masked in place); sampled[k] = the tokens sequence k has produced
    since the start-of-transcript sequence (host lists: the beam bookkeeping lives there anyway)."""
    tb, eot = tok.timestamp_begin, tok.eot
    logits[:, tok.no_timestamps] = NEG_INF                       # <|notimestamps|> is never sampled
    for k, seq in enumerate(sampled):
        last_ts = len(seq) >= 1 and seq[-1] >= tb
        penult_ts = len(seq) < 2 or seq[-2] >= tb
        if last_ts:
            if penult_ts:
                logits[k, tb:] = NEG_INF                          # timestamps come in pairs: a third in a row is not allowed
            else:
                logits[k, :eot] = NEG_INF                         # an opened pair must be closed (or the text must end)
        stamps = [t for t in seq if t >= tb]
        if stamps:
            # timestamps do not decrease; a segment has a non-zero length unless the pair is just being closed
            last = stamps[-1] if (last_ts and not penult_ts) else stamps[-1] + 1
            logits[k, tb:last] = NEG_INF
    if all(len(seq) == 0 for seq in sampled):
        logits[:, :tb] = NEG_INF                                  # the first sampled token is a timestamp ...
        if max_initial_timestamp_index is not None:
            logits[:, tb + max_initial_timestamp_index + 1:] = NEG_INF      # ... no later than max_initial_timestamp
    # if the probability mass on timestamps exceeds every single text token's, a timestamp is sampled
    top_text, _, lse_text = ops.topk_rows(logits[:, :tb], 1)
    _, _, lse_ts = ops.topk_rows(logits[:, tb:], 1)
    force = (lse_ts > top_text[:, 0]).cpu().tolist()              # (the common log-normaliser cancels: compare lse_ts with max text logit)
    del lse_text
    for k, f in enumerate(force):
        if f:
            logits[k, :tb] = NEG_INF


# ------------------------------------------------------------------------------------------------------------------ one window
class _Decoder:
    """DecodingTask for the windows of ONE decode() call: prompt assembly, the filters, greedy / sampling / beam search."""

    def __init__(self, eng, tok: TokenizerSpec, options: DecodingOptions, n_text_ctx: int, rng: Optional[torch.Generator]):
        self.eng, self.tok, self.o = eng, tok, options
        self.n_ctx = n_text_ctx
        self.sample_len = options.sample_len or n_text_ctx // 2
        if options.beam_size is not None and options.best_of is not None:
            raise ValueError("beam_size and best_of can't be given together")
        if options.temperature == 0 and options.best_of is not None:
            raise ValueError("best_of with greedy sampling (T=0) is not compatible")
        if options.patience is not None and options.beam_size is None:
            raise ValueError("patience requires beam_size to be given")
        self.n_group = options.beam_size or options.best_of or 1
        sot_seq = list(tok.sot_sequence(options.language, options.task))
        if options.without_timestamps:
            sot_seq.append(tok.no_timestamps)
        tokens = list(sot_seq)
        if options.prefix:
            prefix = list(options.prefix)
            tokens += prefix[-(self.n_ctx // 2 - self.sample_len):] if self.sample_len < self.n_ctx // 2 else prefix
        if options.prompt:
            tokens = [tok.sot_prev] + list(options.prompt)[-(self.n_ctx // 2 - 1):] + tokens
        self.initial = tokens
        self.sample_begin = len(tokens)
        self.sot_index = tokens.index(tok.sot)
        self.suppress = suppressed_token_ids(options, tok)
        self.max_initial_ts = None
        if options.max_initial_timestamp is not None:
            self.max_initial_ts = round(options.max_initial_timestamp / TIME_PRECISION)
        self.rng = rng

    def _filter(self, logits: torch.Tensor, sampled: Sequence[Sequence[int]]) -> None:
        if self.o.suppress_blank:
            suppress_blank_(logits, len(sampled[0]), self.tok)
        if self.suppress:
            logits[:, self.suppress] = NEG_INF
        if not self.o.without_timestamps:
            apply_timestamp_rules_(logits, sampled, self.tok, self.max_initial_ts)

    @torch.no_grad()
    def run(self, xa: torch.Tensor, n_audio: int) -> List[DecodingResult]:
        eng, tok, G = self.eng, self.tok, self.n_group
        B = xa.shape[0] // n_audio
        N = B * G
        dev = eng.device
        n0 = self.sample_begin
        max_new = min(self.sample_len, self.n_ctx - n0)
        n_max, kv_cross, caches = eng._decode_setup(B, n0, max_new, xa, G)
        seqs = [list(self.initial) for _ in range(N)]
        sum_lp = [0.0] * N
        done = [False] * N
        no_speech = [float("nan")] * B
        beam = self.o.beam_size
        finished: List[Dict[tuple, float]] = [dict() for _ in range(B)]
        max_cand = max(1, round((beam or 1) * (self.o.patience or 1.0)))
        cur = torch.tensor([s[0] for s in seqs], dtype=torch.int64, device=dev)
        for t in range(n_max - 1):
            x = eng._token_step(cur, t, caches, kv_cross, B, G, n_max, n_audio)
            if t == self.sot_index:                                # no-speech probability: softmax at the <|startoftranscript|> position
                lg = eng._last_logits(x[::G].contiguous())
                _, _, lse = ops.topk_rows(lg, 1)
                no_speech = torch.exp(lg[:, tok.no_speech] - lse).cpu().tolist()
            if t + 1 < n0:
                cur = torch.tensor([s[t + 1] for s in seqs], dtype=torch.int64, device=dev)
                continue
            logits = eng._last_logits(x)
            sampled = [s[n0:] for s in seqs]
            self._filter(logits, sampled)
            if beam is None:
                nxt, lp = self._sample(logits)
                for k in range(N):
                    if done[k]:
                        seqs[k].append(tok.eot)
                        continue
                    sum_lp[k] += lp[k]
                    seqs[k].append(nxt[k])
                    done[k] = nxt[k] == tok.eot
                if all(done):
                    break
            else:
                vals, idx, lse = ops.topk_rows(logits, beam + 1)
                lpv = (vals - lse[:, None]).cpu().tolist()
                ix = idx.cpu().tolist()
                new_seqs, new_lp, src = [], [], []
                for i in range(B):
                    scores, sources = {}, {}
                    for j in range(beam):
                        r = i * beam + j
                        for c in range(beam + 1):
                            key = tuple(seqs[r] + [ix[r][c]])
                            scores[key] = sum_lp[r] + lpv[r][c]
                            sources[key] = r
                    saved = 0
                    for key in sorted(scores, key=scores.get, reverse=True):
                        if key[-1] == tok.eot:
                            if len(finished[i]) < max_cand:          # whisper's BeamSearchDecoder.update: first in stays, never evicted
                                finished[i][key] = scores[key]
                        else:
                            new_seqs.append(list(key)); new_lp.append(scores[key]); src.append(sources[key])
                            saved += 1
                            if saved == beam:
                                break
                    while saved < beam:
                        new_seqs.append(list(new_seqs[-1]) if saved else seqs[i * beam] + [tok.eot])
                        new_lp.append(new_lp[-1] if saved else NEG_INF); src.append(src[-1] if saved else i * beam)
                        saved += 1
                seqs, sum_lp = new_seqs, new_lp
                src_dev = torch.tensor(src, dtype=torch.int64, device=dev)
                for li in range(len(caches)):
                    kc = caches[li].view(N, n_max, -1)
                    caches[li] = kc.index_select(0, src_dev).view(N * n_max, -1)
                if all(len(f) >= max_cand for f in finished):
                    break
            if t + 2 >= n_max:
                break
            cur = torch.tensor([s[-1] for s in seqs], dtype=torch.int64, device=dev)
        # ---- finalize + rank (MaximumLikelihoodRanker without length penalty: summed log-probability / length) ----
        results = []
        for i in range(B):
            if beam is None:
                cands = {tuple(seqs[i * G + j] if seqs[i * G + j][-1] == tok.eot else seqs[i * G + j] + [tok.eot]): sum_lp[i * G + j] for j in range(G)}
            else:
                cands = dict(finished[i])
                if len(cands) < beam:                                # not enough finished sequences: the live beams count as ended here
                    for j in sorted(range(beam), key=lambda j_: sum_lp[i * beam + j_], reverse=True):
                        if len(cands) >= beam:
                            break
                        cands[tuple(seqs[i * beam + j] + [tok.eot])] = sum_lp[i * beam + j]
            def body(k_):
                out = list(k_[n0:])
                return out[: out.index(tok.eot)] if tok.eot in out else out
            best = max(cands, key=lambda k_: cands[k_] / max(1, len(body(k_))))
            toks = body(best)
            text = tok.decode(toks)
            results.append(DecodingResult(tokens=toks, text=text.strip() if text is not None else None,
                                          avg_logprob=cands[best] / (len(toks) + 1), no_speech_prob=float(no_speech[i]),
                                          temperature=float(self.o.temperature),
                                          compression_ratio=compression_ratio(text if text is not None else toks), language=self.o.language))
        return results

    def _sample(self, logits: torch.Tensor) -> Tuple[List[int], List[float]]:
        """GreedyDecoder.update: argmax at T = 0, else one draw from softmax(logits / T) (Gumbel-max over the scaled logits)."""
        _, _, lse = ops.topk_rows(logits, 1)
        if self.o.temperature == 0:
            nxt = ops.argmax_rows(logits)
        else:
            u = torch.rand(logits.shape, device=logits.device, generator=self.rng).clamp_(1e-20, 1.0)
            nxt = ops.argmax_rows((logits / float(self.o.temperature) - torch.log(-torch.log(u))).contiguous())
        lp = logits.gather(1, nxt[:, None])[:, 0] - lse
        return nxt.cpu().tolist(), lp.cpu().tolist()


def resolve_tokenizer(model, tokenizer=None, language: Optional[str] = None, task: Optional[str] = None) -> TokenizerSpec:
    """What decode / transcribe work with.  None: lyricalignment_amd.tokenizer.get_tokenizer for the model's vocabulary (special
    ids always; text and the non-speech suppression set when whisper's rank file can be found -- tokenizer.py); a
    lyricalignment_amd.tokenizer.Tokenizer or an openai-whisper Tokenizer (anything with .decode and .non_speech_tokens): wrapped;
    a TokenizerSpec: as it is."""
    if isinstance(tokenizer, TokenizerSpec):
        return tokenizer
    multilingual = int(getattr(model.dims, "n_vocab", 51865)) >= 51865          # whisper's Whisper.is_multilingual
    if tokenizer is None:
        from .tokenizer import get_tokenizer
        return get_tokenizer(multilingual, language=language if multilingual else None, task=task if multilingual else None).spec()
    if hasattr(tokenizer, "spec"):
        return tokenizer.spec()
    blank = tokenizer.encode(" ") if hasattr(tokenizer, "encode") else [220]
    return TokenizerSpec(multilingual=multilingual, codec=tokenizer, non_speech_ids=tuple(getattr(tokenizer, "non_speech_tokens", ())),
                         blank_id=int(blank[0]) if len(blank) == 1 else 220)


@torch.no_grad()
def decode(model, audio_features: torch.Tensor, options: Optional[DecodingOptions] = None, tokenizer: Optional[TokenizerSpec] = None,
           rng: Optional[torch.Generator] = None) -> List[DecodingResult]:
    """whisper.decode for encoder outputs: audio_features [B, n_audio, d] -> one DecodingResult per window."""
    from .module.align_model import decoder_engine_of
    eng = decoder_engine_of(model)
    tok = resolve_tokenizer(model, tokenizer, (options or DecodingOptions()).language, (options or DecodingOptions()).task)
    B, n_audio, d = audio_features.shape
    xa = audio_features.to(device=eng.device, dtype=torch.float32).contiguous().view(B * n_audio, d)
    return _Decoder(eng, tok, options or DecodingOptions(), int(model.dims.n_text_ctx), rng).run(xa, n_audio)


# ------------------------------------------------------------------------------------------------------------------ transcribe
@torch.no_grad()
def transcribe(model, audio, *, task: str = "transcribe", language: Optional[str] = "en", temperature=(0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
               compression_ratio_threshold: Optional[float] = 2.4, logprob_threshold: Optional[float] = -1.0,
               no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
               initial_prompt: Optional[Sequence[int]] = None, beam_size: Optional[int] = None, best_of: Optional[int] = None,
               patience: Optional[float] = None, without_timestamps: bool = False, suppress_tokens="-1", suppress_blank: bool = True,
               max_initial_timestamp: Optional[float] = 1.0, tokenizer: Optional[TokenizerSpec] = None,
               rng: Optional[torch.Generator] = None, mel: Optional[torch.Tensor] = None) -> dict:
    """whisper.transcribe for one recording: audio float32 [N] at 16 kHz (or a ready log-mel [80, frames]).
    -> {"text", "tokens", "segments": [{seek, start, end, tokens, text, temperature, avg_logprob, compression_ratio, no_speech_prob}],
        "language"}.  `initial_prompt`: token ids, or text when the vocabulary is available (tokenizer.py)."""
    from .audio_frontend import log_mel_spectrogram
    from .module.align_model import decoder_engine_of
    eng = decoder_engine_of(model)
    tok = resolve_tokenizer(model, tokenizer, language, task)
    if isinstance(initial_prompt, str):                               # whisper takes text here: needs the vocabulary
        if tok.codec is None or not hasattr(tok.codec, "encode"):
            raise FileNotFoundError("initial_prompt as text needs whisper's vocabulary (lyricalignment_amd/tokenizer.py); pass token ids")
        initial_prompt = tok.codec.encode(" " + initial_prompt.strip())
    if mel is None:
        mel = log_mel_spectrogram(np.asarray(audio, dtype=np.float32), device=str(eng.device))
    mel = mel.to(eng.device)
    content_frames = int(mel.shape[-1])
    mel = torch.nn.functional.pad(mel, (0, N_FRAMES))              # whisper pads 30 s of silence behind the recording
    temps = tuple(temperature) if isinstance(temperature, (list, tuple)) else (float(temperature),)
    n_text_ctx = int(model.dims.n_text_ctx)
    all_tokens: List[int] = list(initial_prompt or [])
    prompt_reset_since = 0
    segments: List[dict] = []
    seek = 0

    def decode_with_fallback(segment_mel: torch.Tensor, prompt: Sequence[int]) -> DecodingResult:
        xa = eng.encode(segment_mel[None], out_dtype=torch.float32).float()
        result = None
        for t in temps:
            o = DecodingOptions(task=task, language=language, temperature=float(t), prompt=list(prompt) or None,
                                suppress_tokens=suppress_tokens, suppress_blank=suppress_blank, without_timestamps=without_timestamps,
                                max_initial_timestamp=max_initial_timestamp)
            if t > 0:
                o.best_of = best_of                                   # beam search only at temperature 0
            else:
                o.beam_size, o.patience = beam_size, patience
            result = _Decoder(eng, tok, o, n_text_ctx, rng).run(xa, xa.shape[0])[0]
            needs_fallback = False
            if compression_ratio_threshold is not None and result.compression_ratio > compression_ratio_threshold:
                needs_fallback = True                                 # too repetitive
            if logprob_threshold is not None and result.avg_logprob < logprob_threshold:
                needs_fallback = True                                 # average log-probability too low
            if no_speech_threshold is not None and result.no_speech_prob > no_speech_threshold:
                needs_fallback = False                                # silence
            if not needs_fallback:
                break
        return result

    def new_segment(start: float, end: float, tokens: Sequence[int], result: DecodingResult) -> dict:
        text_tokens = [t for t in tokens if t < tok.eot]
        return {"seek": seek, "start": start, "end": end, "tokens": list(tokens), "text": tok.decode(text_tokens),
                "temperature": result.temperature, "avg_logprob": result.avg_logprob, "compression_ratio": result.compression_ratio,
                "no_speech_prob": result.no_speech_prob}

    while seek < content_frames:
        time_offset = seek * HOP_LENGTH / SAMPLE_RATE
        segment_size = min(N_FRAMES, content_frames - seek)
        segment_mel = pad_or_trim(mel[:, seek: seek + segment_size], N_FRAMES)
        segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE
        result = decode_with_fallback(segment_mel, all_tokens[prompt_reset_since:])
        tokens = list(result.tokens)
        if no_speech_threshold is not None:
            should_skip = result.no_speech_prob > no_speech_threshold
            if logprob_threshold is not None and result.avg_logprob > logprob_threshold:
                should_skip = False                                   # the text is plausible: do not skip
            if should_skip:
                seek += segment_size
                continue
        current: List[dict] = []
        is_ts = [t >= tok.timestamp_begin for t in tokens]
        single_ending = is_ts[-2:] == [False, True]
        consecutive = [i for i in range(1, len(tokens)) if is_ts[i] and is_ts[i - 1]]
        if consecutive:                                               # output contains two consecutive timestamp tokens: slice there
            slices = list(consecutive)
            if single_ending:
                slices.append(len(tokens))
            last_slice = 0
            for cur_slice in slices:
                sl = tokens[last_slice:cur_slice]
                start_pos, end_pos = sl[0] - tok.timestamp_begin, sl[-1] - tok.timestamp_begin
                current.append(new_segment(time_offset + start_pos * TIME_PRECISION, time_offset + end_pos * TIME_PRECISION, sl, result))
                last_slice = cur_slice
            if single_ending:
                seek += segment_size                                  # a single timestamp at the end: no speech after it
            else:
                seek += (tokens[last_slice - 1] - tok.timestamp_begin) * INPUT_STRIDE      # resume at the last closed timestamp
        else:
            duration = segment_duration
            stamps = [t for t in tokens if t >= tok.timestamp_begin]
            if stamps and stamps[-1] != tok.timestamp_begin:
                duration = (stamps[-1] - tok.timestamp_begin) * TIME_PRECISION
            current.append(new_segment(time_offset, time_offset + duration, tokens, result))
            seek += segment_size
        for seg in current:                                           # drop segments that are empty or of zero length
            if seg["start"] == seg["end"] or not [t for t in seg["tokens"] if t < tok.eot]:
                seg["tokens"], seg["text"] = [], ("" if tok.codec is not None else None)
        segments += [s for s in current if s["tokens"]]
        all_tokens += [t for s in current for t in s["tokens"]]
        if not condition_on_previous_text or result.temperature > 0.5:
            prompt_reset_since = len(all_tokens)                      # AFTER this window's tokens went in: do not feed a
                                                                      # probably wrong window forward as a prompt
    body = all_tokens[len(initial_prompt or []):]
    return {"text": tok.decode([t for t in body if t < tok.eot]), "tokens": body, "segments": segments, "language": language}
