"""Tokenizers of the text towers -- host code; the towers themselves are `encoders/text.py`.

The reference obtains them from the un-vendored model packages: `open_clip.get_tokenizer(card)` (clip_utils.py:80) for the
CLIP / SigLIP cards and `transforms.get_text_tokenizer(model.context_length)` of perception_models (clip_utils.py:110) for
PE, and calls `self.tokenizer(phrase)` once per phrase (clip_generator.py:170).  Neither package, nor their vocabulary
files, exist offline, so the two algorithms are written out here and take the vocabulary as a *path*:

  * `SimpleTokenizer(bpe_path, context_length)`: CLIP's lower-cased byte-level BPE (OpenAI CLIP `simple_tokenizer.py`, the
    same file open_clip and perception_models ship: `bpe_simple_vocab_16e6.txt.gz`): context 77 (CLIP) / 32 (PE), ids
    `<start_of_text>` = V-2, `<end_of_text>` = V-1 (the highest id: what the tower's argmax pooling relies on), zero padding,
    truncation keeps `<end_of_text>` in the last slot.
  * `SigLIPTokenizer(spm_path, context_length)`: SigLIP's SentencePiece vocabulary with open_clip's "canonicalize" cleaning
    (punctuation removed, lower-cased), `</s>` = 1 appended, padded with the same id to the context length (64; 16 for
    SigLIP-224).  SigLIP2's Gemma vocabulary needs its own file and is loaded the same way (lower-casing only).

Parity: pinned in tests/test_tokenizer.py against HuggingFace transformers' CLIPTokenizer (Rust `tokenizers` backend) and
SiglipTokenizer on vocabularies built inside the test (a BPE merge list trained on a small corpus; a SentencePiece model
trained on the same corpus), i.e. against independent implementations of the same algorithms.
"""
from __future__ import annotations

import gzip
import html
import string
from functools import lru_cache
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import torch


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """The byte -> printable unicode character table of GPT-2 / CLIP: printable latin-1 bytes map to themselves, the other
    68 bytes to code points 256.."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    chars = keep[:]
    extra = 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + extra)
            extra += 1
    return {b: chr(c) for b, c in zip(keep, chars)}


def read_merges(bpe_path: str, vocab_size: int = 49408) -> List[Tuple[str, str]]:
    """Merge rules of a CLIP BPE file (plain or .gz): line 0 is a header, then one `left right` pair per line; the model
    keeps the first vocab_size - 512 - 2 of them."""
    opener = gzip.open if str(bpe_path).endswith(".gz") else open
    with opener(bpe_path, "rb") as f:
        lines = f.read().decode("utf-8").split("\n")
    lines = lines[1:vocab_size - 512 - 2 + 1]                  # upstream: merges[1:49152-256-2+1] with 49152 = 49408 - 256
    return [tuple(l.split()) for l in lines if l.strip()]


def _whitespace_clean(text: str) -> str:
    return " ".join(text.split()).strip()


def _basic_clean(text: str) -> str:
    # upstream runs ftfy.fix_text first (mojibake repair; a no-op on clean text and not installable here)
    return html.unescape(html.unescape(text)).strip()


class SimpleTokenizer:
    """CLIP BPE.  `tok(texts)` -> i64 [n, context_length] like open_clip's tokenizer object."""

    def __init__(self, bpe_path: Union[str, Sequence[Tuple[str, str]]], context_length: int = 77, vocab_size: int = 49408):
        import regex
        merges = read_merges(bpe_path, vocab_size) if isinstance(bpe_path, str) else [tuple(m) for m in bpe_path]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]
        self.context_length = context_length
        self._cache: Dict[str, Tuple[str, ...]] = {}
        self._pat = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)

    @property
    def vocab_size(self) -> int:
        return len(self.encoder)

    def _bpe(self, token: str) -> Tuple[str, ...]:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best, at = None, -1
            for i in range(len(word) - 1):                      # lowest-ranked adjacent pair
                r = self.ranks.get((word[i], word[i + 1]))
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            a, b = word[at], word[at + 1]
            out, i = [], 0
            while i < len(word):                                # merge every occurrence of that pair, left to right
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        res = tuple(word)
        self._cache[token] = res
        return res

    def encode(self, text: str) -> List[int]:
        text = _whitespace_clean(_basic_clean(text)).lower()
        ids: List[int] = []
        for piece in self._pat.findall(text):
            if piece in ("<start_of_text>", "<end_of_text>"):
                ids.append(self.encoder[piece])
                continue
            mapped = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(mapped))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        inv = {c: b for b, c in self.byte_encoder.items()}
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = 0) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        n = context_length or self.context_length
        out = torch.zeros((len(texts), n), dtype=torch.long)
        for r, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > n:
                ids = ids[:n]
                ids[-1] = self.eot
            out[r, :len(ids)] = torch.tensor(ids)
        return out


def canonicalize_text(text: str, keep_punctuation_exact_string: str = "") -> str:
    """open_clip's `canonicalize_text` (the cleaning its SigLIP tokenizer configs select): underscores -> spaces, ASCII
    punctuation removed, lower-cased, whitespace collapsed."""
    text = text.replace("_", " ")
    table = str.maketrans("", "", string.punctuation)
    if keep_punctuation_exact_string:
        text = keep_punctuation_exact_string.join(p.translate(table) for p in text.split(keep_punctuation_exact_string))
    else:
        text = text.translate(table)
    return " ".join(text.lower().split()).strip()


class SigLIPTokenizer:
    """SentencePiece tokenizer of the SigLIP text towers: canonicalise, encode, append `</s>`, pad with it."""

    def __init__(self, spm_path: str, context_length: int = 64, canonicalize: bool = True):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor(model_file=spm_path)
        self.context_length, self.canonicalize = context_length, canonicalize
        eos = self.sp.piece_to_id("</s>")
        self.eos = eos if eos >= 0 and self.sp.id_to_piece(eos) == "</s>" else 1
        self.pad = self.eos

    @property
    def vocab_size(self) -> int:
        return self.sp.get_piece_size()

    def encode(self, text: str) -> List[int]:
        text = canonicalize_text(text) if self.canonicalize else " ".join(text.lower().split())
        return list(self.sp.encode(text)) + [self.eos]

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = 0) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        n = context_length or self.context_length
        out = torch.full((len(texts), n), self.pad, dtype=torch.long)
        for r, t in enumerate(texts):
            ids = self.encode(t)
            if len(ids) > n:                                    # truncation keeps the closing </s> (HF truncation=True)
                ids = ids[:n - 1] + [self.eos]
            out[r, :len(ids)] = torch.tensor(ids)
        return out


def get_tokenizer(model_card: str, vocab_path: str):
    """`open_clip.get_tokenizer(card)` / `transforms.get_text_tokenizer(context)` for the cards of clip_utils.py:53-75."""
    from .text import SPECS
    alias = {"PE-Core-L-14-336": "PE-Core-L14-336", "ViT-H-14-qg": "ViT-H-14", "ViT-H-14-378qg": "ViT-H-14"}
    spec = SPECS[alias.get(model_card, model_card)]
    if model_card.startswith("SigLIP"):
        return SigLIPTokenizer(vocab_path, spec.context, canonicalize=not model_card.startswith("SigLIP2"))
    return SimpleTokenizer(vocab_path, spec.context, spec.vocab)
