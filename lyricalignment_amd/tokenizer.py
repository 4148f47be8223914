"""Stand-in for `whisper.tokenizer` (openai-whisper is not installed here; the reference imports `get_tokenizer` / `Tokenizer`
from it: inference_alignment.py:17,195, dataset.py:9,45-100, train_multitask.py).

Two parts:
* the SPECIAL tokens -- ids and names are constants of the two published vocabularies (transcribe.TokenizerSpec) and need no
  asset: `tokenizer.sot`, `.eot`, `.no_speech`, `.no_timestamps`, `.timestamp_begin`, `.special_tokens["<|zh|>"]`,
  `.sot_sequence` ... work out of the box;
* the TEXT part -- a byte-pair codec over whisper's rank files (`multilingual.tiktoken` / `gpt2.tiktoken`: one
  `base64(token bytes) rank` pair per line, 50257 / 50256 entries).  The files are assets of openai-whisper and are not in
  this image (no network): `get_tokenizer(..., vocab_path=...)`, `$LA_WHISPER_ASSETS/<name>.tiktoken`, or an installed
  openai-whisper's own assets directory supply them; without one `encode` / `decode` raise FileNotFoundError naming the file.
  The codec itself (GPT-2 pre-tokenisation pattern, lowest-rank-first pair merging) is written here and held to an
  independent implementation (HF `tokenizers` byte-level BPE) on a synthetic vocabulary in tests/test_host_logic.py.
"""
from __future__ import annotations

import base64
import os
from functools import cached_property, lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

from .transcribe import LANGUAGES, TokenizerSpec

# GPT-2's pre-tokenisation pattern (the one both whisper rank files are used with): contractions, letter runs, digit runs,
# other runs -- each with an optional leading space -- and whitespace.
_PRETOKENIZE = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


class BytePairCodec:
    """Byte-level byte-pair encoding over a rank table {token bytes: rank} whose ranks 0..n-1 are the token ids."""

    def __init__(self, ranks: Dict[bytes, int]):
        import regex
        if sorted(ranks.values()) != list(range(len(ranks))):
            raise ValueError("byte-pair ranks must be exactly 0 .. n-1")
        missing = [b for b in range(256) if bytes([b]) not in ranks]
        if missing:
            raise ValueError(f"byte-pair vocabulary lacks single-byte tokens {missing[:4]}...")
        self.ranks = ranks
        self.tokens: List[bytes] = [b""] * len(ranks)
        for tok, r in ranks.items():
            self.tokens[r] = tok
        self._split = regex.compile(_PRETOKENIZE)
        self._cache: Dict[bytes, Tuple[int, ...]] = {}

    @classmethod
    def from_tiktoken_file(cls, path: str) -> "BytePairCodec":
        ranks: Dict[bytes, int] = {}
        with open(path, "rb") as f:
            for line in f:
                if line.strip():
                    tok, rank = line.split()
                    ranks[base64.b64decode(tok)] = int(rank)
        return cls(ranks)

    @property
    def n_vocab(self) -> int:
        return len(self.tokens)

    def _merge(self, piece: bytes) -> Tuple[int, ...]:
        """One pre-token -> ids: start from single bytes; repeatedly join the adjacent pair whose concatenation has the lowest
        rank (leftmost on ties) until no adjacent pair is in the table."""
        hit = self._cache.get(piece)
        if hit is not None:
            return hit
        parts = [bytes([b]) for b in piece]
        ranks = self.ranks
        while len(parts) > 1:
            best, at = None, -1
            for i in range(len(parts) - 1):
                r = ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            parts[at:at + 2] = [parts[at] + parts[at + 1]]
        out = tuple(ranks[p] for p in parts)
        if len(self._cache) < 1 << 16:
            self._cache[piece] = out
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for m in self._split.finditer(text):
            ids.extend(self._merge(m.group().encode("utf-8")))
        return ids

    def decode_bytes(self, ids: Iterable[int]) -> bytes:
        return b"".join(self.tokens[int(i)] for i in ids)

    def decode(self, ids: Iterable[int]) -> str:
        return self.decode_bytes(ids).decode("utf-8", errors="replace")


def _special_names(num_languages: int = len(LANGUAGES)) -> List[str]:
    """The special tokens in id order, starting at the rank file's size (whisper/tokenizer.py get_encoding)."""
    return (["<|endoftext|>", "<|startoftranscript|>"] + [f"<|{lang}|>" for lang in LANGUAGES[:num_languages]]
            + ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
            + [f"<|{i * 0.02:.2f}|>" for i in range(1501)])


def _find_vocab(name: str, vocab_path: Optional[str]) -> Optional[str]:
    cands = []
    if vocab_path:
        cands.append(vocab_path if not os.path.isdir(vocab_path) else os.path.join(vocab_path, f"{name}.tiktoken"))
    if os.environ.get("LA_WHISPER_ASSETS"):
        cands.append(os.path.join(os.environ["LA_WHISPER_ASSETS"], f"{name}.tiktoken"))
    try:                                    # an installed openai-whisper's own assets (absent in this image)
        import importlib.util
        spec = importlib.util.find_spec("whisper")
        if spec is not None and spec.origin and "lyricalignment_amd" not in spec.origin:
            cands.append(os.path.join(os.path.dirname(spec.origin), "assets", f"{name}.tiktoken"))
    except (ImportError, ValueError):
        pass
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


class Tokenizer:
    """whisper.tokenizer.Tokenizer's surface: special-token properties, `sot_sequence`, `encode` / `decode` /
    `decode_with_timestamps`, `non_speech_tokens`, language helpers."""

    def __init__(self, multilingual: bool = True, language: Optional[str] = None, task: Optional[str] = None,
                 codec: Optional[BytePairCodec] = None, vocab_name: str = "multilingual", num_languages: int = len(LANGUAGES)):
        self.multilingual = multilingual
        self.language, self.task = language, task
        self.num_languages = num_languages
        self.encoding = codec               # BytePairCodec or None (special tokens only)
        self._vocab_name = vocab_name
        self._spec = TokenizerSpec(multilingual=multilingual)
        base = self._spec.eot
        if codec is not None and codec.n_vocab != base:
            raise ValueError(f"the {vocab_name} vocabulary has {base} byte-pair tokens, the file given has {codec.n_vocab}")
        self.special_tokens: Dict[str, int] = {n: base + i for i, n in enumerate(_special_names(num_languages))}
        self._special_by_id = {i: n for n, i in self.special_tokens.items()}
        sot = [self.sot]
        if language is not None:
            sot.append(self.to_language_token(language))
        if task is not None:
            sot.append(self.transcribe if task == "transcribe" else self.translate)
        self.sot_sequence: Tuple[int, ...] = tuple(sot)

    # ---- special tokens (whisper/tokenizer.py property names) ---------------------------------------------------------
    eot = property(lambda self: self.special_tokens["<|endoftext|>"])
    sot = property(lambda self: self.special_tokens["<|startoftranscript|>"])
    translate = property(lambda self: self.special_tokens["<|translate|>"])
    transcribe = property(lambda self: self.special_tokens["<|transcribe|>"])
    sot_lm = property(lambda self: self.special_tokens["<|startoflm|>"])
    sot_prev = property(lambda self: self.special_tokens["<|startofprev|>"])
    no_speech = property(lambda self: self.special_tokens["<|nospeech|>"])
    no_timestamps = property(lambda self: self.special_tokens["<|notimestamps|>"])
    timestamp_begin = property(lambda self: self.special_tokens["<|0.00|>"])

    @property
    def n_vocab(self) -> int:
        return self.eot + len(self.special_tokens)

    @property
    def language_token(self) -> int:
        if self.language is None:
            raise ValueError("This tokenizer does not have language token configured")
        return self.to_language_token(self.language)

    def to_language_token(self, language: str) -> int:
        tok = self.special_tokens.get(f"<|{language}|>")
        if tok is None:
            raise KeyError(f"Language {language} not found in tokenizer.")
        return tok

    @property
    def all_language_tokens(self) -> Tuple[int, ...]:
        return tuple(self.special_tokens[f"<|{lang}|>"] for lang in LANGUAGES[:self.num_languages])

    @property
    def all_language_codes(self) -> Tuple[str, ...]:
        return tuple(LANGUAGES[:self.num_languages])

    @property
    def sot_sequence_including_notimestamps(self) -> Tuple[int, ...]:
        return tuple(list(self.sot_sequence) + [self.no_timestamps])

    # ---- text ---------------------------------------------------------------------------------------------------------
    def _codec(self) -> BytePairCodec:
        if self.encoding is None:
            raise FileNotFoundError(
                f"whisper's byte-pair vocabulary '{self._vocab_name}.tiktoken' is not available (it ships with openai-whisper, "
                "which is not installed): pass vocab_path= to get_tokenizer or set LA_WHISPER_ASSETS to the directory holding it")
        return self.encoding

    def encode(self, text: str, **_) -> List[int]:
        return self._codec().encode(text)

    def decode(self, token_ids: Sequence[int], **_) -> str:
        """Text of the ids below the timestamp range; special tokens among them read as their names (tiktoken's decode)."""
        codec = self._codec()
        out, run = [], []
        for t in (int(t) for t in token_ids):
            if t >= self.timestamp_begin:
                continue
            if t >= self.eot:
                out.append(codec.decode_bytes(run)); run = []
                out.append(self._special_by_id[t].encode())
            else:
                run.append(t)
        out.append(codec.decode_bytes(run))
        return b"".join(out).decode("utf-8", errors="replace")

    def decode_with_timestamps(self, token_ids: Sequence[int], **_) -> str:
        """Timestamp tokens annotated as <|1.08|>, everything else as decode() gives it."""
        codec = self._codec()
        out, run = [], []
        for t in (int(t) for t in token_ids):
            if t >= self.eot:
                out.append(codec.decode_bytes(run)); run = []
                out.append((f"<|{(t - self.timestamp_begin) * 0.02:.2f}|>" if t >= self.timestamp_begin else self._special_by_id[t]).encode())
            else:
                run.append(t)
        out.append(codec.decode_bytes(run))
        return b"".join(out).decode("utf-8", errors="replace")

    @cached_property
    def non_speech_tokens(self) -> Tuple[int, ...]:
        """The ids whisper suppresses so that speaker tags / non-speech annotations ("[laughter]", "♪♪♪") are not sampled:
        the listed symbols on their own and after a space, when they are ONE token (the musical symbols also when they are not:
        their first token), plus " -" and " '".  Same construction as whisper/tokenizer.py non_speech_tokens."""
        symbols = list("\"#()*+/:;<=>@[\\]^_`{|}~「」『』")
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        miscellaneous = set("♩♪♫♬♭♮♯")
        codec = self._codec()
        result = {codec.encode(" -")[0], codec.encode(" '")[0]}
        for symbol in symbols + sorted(miscellaneous):
            for tokens in (codec.encode(symbol), codec.encode(" " + symbol)):
                if len(tokens) == 1 or symbol in miscellaneous:
                    result.add(tokens[0])
        return tuple(sorted(result))

    def spec(self) -> TokenizerSpec:
        """The view transcribe.decode / transcribe take: ids always; text and the non-speech set when the vocabulary is there."""
        if self.encoding is None:
            return TokenizerSpec(multilingual=self.multilingual)
        blank = self.encoding.encode(" ")
        return TokenizerSpec(multilingual=self.multilingual, codec=self, non_speech_ids=self.non_speech_tokens,
                             blank_id=blank[0] if len(blank) == 1 else 220)


@lru_cache(maxsize=None)
def _codec_from(path: str) -> BytePairCodec:
    return BytePairCodec.from_tiktoken_file(path)


def get_tokenizer(multilingual: bool = True, *, num_languages: int = len(LANGUAGES), language: Optional[str] = None,
                  task: Optional[str] = None, vocab_path: Optional[str] = None) -> Tokenizer:
    """whisper.tokenizer.get_tokenizer: English-only models get neither language nor task; multilingual ones default to
    language 'en', task 'transcribe'.  Language names ("chinese") are accepted next to codes ("zh")."""
    if language is not None:
        language = language.lower()
        if language not in LANGUAGES:
            from .transcribe import LANGUAGE_NAMES
            if language in LANGUAGE_NAMES:
                language = LANGUAGE_NAMES[language]
            else:
                raise ValueError(f"Unsupported language: {language}")
    if multilingual:
        name, language, task = "multilingual", language or "en", task or "transcribe"
    else:
        name, language, task = "gpt2", None, None
    path = _find_vocab(name, vocab_path)
    return Tokenizer(multilingual=multilingual, language=language, task=task, codec=_codec_from(path) if path else None,
                     vocab_name=name, num_languages=num_languages)
