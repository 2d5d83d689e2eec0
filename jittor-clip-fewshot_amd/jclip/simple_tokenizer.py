"""Byte-level BPE tokenizer for CLIP text prompts (host side, integer path).

Mirrors the interface of the reference's ``jclip/simple_tokenizer.py:67-149``
(``SimpleTokenizer`` with ``encoder`` / ``decoder`` / ``encode`` / ``decode``)
so ``jclip.clip.tokenize`` is a drop-in.  Implementation is this repo's own:
a rank-table driven merge loop over a symbol list, with a per-word cache.

Differences from the reference, both deliberate:
  * the vocabulary file: the reference looks for ``bpe_simple_vocab_16e6.txt.gz``
    (``simple_tokenizer.py:12-13``) while the file it ships is named
    ``bpe_simple_vocab_16e6.txt`` but holds gzip bytes.  ``default_bpe()`` resolves, in
    order, ``$CLIPFS_BPE_PATH``, then either name next to this module (the vocabulary
    ships in-tree); gzip is detected from the magic bytes, not the suffix.
  * ``ftfy.fix_text`` (``simple_tokenizer.py:55``) is applied when ``ftfy`` is
    importable and skipped otherwise (identity on the ASCII class names and
    templates this task uses).
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Dict, Iterable, List, Tuple

import regex

try:  # optional, not installed in the build image
    import ftfy as _ftfy
except Exception:  # pragma: no cover
    _ftfy = None

SOT = "<|startoftext|>"
EOT = "<|endoftext|>"
_N_MERGES = 49152 - 256 - 2  # vocabulary 49408 = 256 bytes * 2 + merges + 2 specials
_WORD_END = "</w>"

_SPLIT = regex.compile(
    r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
    regex.IGNORECASE)
_WS = regex.compile(r"\s+")


@lru_cache()
def default_bpe() -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    names = ("bpe_simple_vocab_16e6.txt.gz", "bpe_simple_vocab_16e6.txt")
    cands = [os.environ.get("CLIPFS_BPE_PATH", "")]
    cands += [os.path.join(here, n) for n in names]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError(
        "CLIP BPE vocabulary not found; set CLIPFS_BPE_PATH or place "
        "bpe_simple_vocab_16e6.txt(.gz) next to jclip/simple_tokenizer.py")


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """Reversible byte -> printable unicode map (GPT-2 alphabet): printable
    latin-1 bytes map to themselves, the other 68 bytes to U+0100.."""
    printable = [*range(0x21, 0x7F), *range(0xA1, 0xAD), *range(0xAE, 0x100)]
    table = {b: chr(b) for b in printable}
    shift = 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + shift)
            shift += 1
    # insertion order must be "printable first, then the rest" -- the vocabulary
    # ids are defined by this order.
    return table


def _read_merges(path: str) -> List[Tuple[str, str]]:
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    lines = raw.decode("utf-8").split("\n")
    return [tuple(ln.split()) for ln in lines[1:_N_MERGES + 1]]  # line 0 is a version header


def basic_clean(text: str) -> str:
    if _ftfy is not None:
        text = _ftfy.fix_text(text)
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text: str) -> str:
    return _WS.sub(" ", text).strip()


class _NativeBpe:
    """The merge loop in native code (csrc/bpe.hip, C ABI ``clipfs_bpe_*``): words in, vocabulary ids out."""

    def __init__(self, merges: List[Tuple[str, str]]):
        import ctypes as C

        import numpy as np
        from clipfs import _lib
        self._np, self._C = np, C
        self._lib = _lib.load()
        text = "\n".join(a + " " + b for a, b in merges).encode("utf-8") + b"\n"
        self._h = self._lib.clipfs_bpe_create(text, len(text), len(merges))
        if not self._h:
            raise RuntimeError("clipfs_bpe_create failed: " + self._lib.clipfs_last_error().decode("utf-8", "replace"))

    def encode_words(self, words: List[bytes]) -> List[List[int]]:
        np = self._np
        if not words:
            return []
        blob = np.frombuffer(b"".join(words), dtype=np.uint8) if any(words) else np.zeros(1, np.uint8)
        offs = np.zeros(len(words) + 1, np.int32)
        np.cumsum([len(w) for w in words], out=offs[1:])
        ids = np.empty(max(int(offs[-1]), 1), np.int32)
        counts = np.empty(len(words), np.int32)
        n = self._lib.clipfs_bpe_encode(self._h, blob.ctypes.data, offs.ctypes.data, len(words), ids.ctypes.data,
                                        counts.ctypes.data, ids.size)
        if n < 0:
            raise RuntimeError("clipfs_bpe_encode failed: " + self._lib.clipfs_last_error().decode("utf-8", "replace"))
        out, pos = [], 0
        for c in counts.tolist():
            out.append(ids[pos:pos + c].tolist())
            pos += c
        return out

    def __del__(self):
        try:
            if self._h:
                self._lib.clipfs_bpe_destroy(self._h)
        except Exception:
            pass


_WARNED_NATIVE = False


class SimpleTokenizer:
    """``native=True`` (default): the merge loop runs in the library's C++ BPE core; ``native=False`` keeps the pure
    Python loop below (the cross-check of tests/test_tokenizer.py).  Both produce the same ids."""

    def __init__(self, bpe_path: str = None, native: bool = True):
        merges = _read_merges(bpe_path or default_bpe())
        self._native = None
        if native:  # a preference, not a requirement: host-side tokenisation must work on a box without the built library
            try:
                self._native = _NativeBpe(merges)
            except (OSError, RuntimeError) as e:  # ClipfsError is a RuntimeError
                global _WARNED_NATIVE
                if not _WARNED_NATIVE:
                    import warnings
                    warnings.warn(f"native BPE core unavailable ({e}); using the Python merge loop (same ids)")
                    _WARNED_NATIVE = True
        self._id_cache: Dict[str, List[int]] = {}
        b2u = bytes_to_unicode()
        self.byte_encoder = b2u
        self.byte_decoder = {u: b for b, u in b2u.items()}
        symbols = list(b2u.values())
        vocab = symbols + [s + _WORD_END for s in symbols] + [a + b for a, b in merges] + [SOT, EOT]
        self.encoder: Dict[str, int] = {tok: i for i, tok in enumerate(vocab)}
        self.decoder: Dict[int, str] = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks: Dict[Tuple[str, str], int] = {m: r for r, m in enumerate(merges)}
        self.cache: Dict[str, str] = {SOT: SOT, EOT: EOT}
        self.pat = _SPLIT

    # -- BPE ---------------------------------------------------------------
    def _merge_word(self, symbols: List[str]) -> List[str]:
        ranks = self.bpe_ranks
        inf = len(ranks)
        while len(symbols) > 1:
            # lowest-rank adjacent pair wins; every occurrence is merged left to right
            best_rank, best = inf, None
            for pair in zip(symbols, symbols[1:]):
                r = ranks.get(pair, inf)
                if r < best_rank:
                    best_rank, best = r, pair
            if best is None:
                break
            a, b = best
            out: List[str] = []
            i, n = 0, len(symbols)
            while i < n:
                if i + 1 < n and symbols[i] == a and symbols[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(symbols[i])
                    i += 1
            symbols = out
        return symbols

    def bpe(self, token: str) -> str:
        hit = self.cache.get(token)
        if hit is not None:
            return hit
        symbols = list(token[:-1]) + [token[-1] + _WORD_END]
        word = " ".join(self._merge_word(symbols))
        self.cache[token] = word
        return word

    # -- public ------------------------------------------------------------
    def encode(self, text: str) -> List[int]:
        text = whitespace_clean(basic_clean(text)).lower()
        ids: List[int] = []
        b2u, enc = self.byte_encoder, self.encoder
        pieces = self.pat.findall(text)
        if self._native is not None:
            cache = self._id_cache
            todo = [p for p in dict.fromkeys(pieces) if p not in cache and p not in (SOT, EOT)]
            if todo:
                for p, got in zip(todo, self._native.encode_words([p.encode("utf-8") for p in todo])):
                    cache[p] = got
            for p in pieces:
                ids.extend([enc[p]] if p in (SOT, EOT) else cache[p])
            return ids
        for piece in pieces:
            mapped = "".join(b2u[b] for b in piece.encode("utf-8"))
            ids.extend(enc[s] for s in self.bpe(mapped).split(" "))
        return ids

    def decode(self, tokens: Iterable[int]) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        raw = bytearray(self.byte_decoder[c] for c in text)
        return raw.decode("utf-8", errors="replace").replace(_WORD_END, " ")
