"""f2 tokenizers (host code, CPU): ovo_amd/encoders/tokenizer.py against independent implementations of the same algorithms --
HuggingFace transformers' CLIPTokenizer (Rust `tokenizers` backend) and SiglipTokenizer (SentencePiece backend) -- on
vocabularies built here (the real vocabulary files are not available offline): a BPE merge list learnt from a small corpus and
a SentencePiece model trained on the same corpus."""
import collections
import os

import pytest
import torch

CORPUS = """a photo of a chair . a photo of the table in a room . there is a lamp next to the sofa .
the kitchen counter has a sink and a microwave oven . a bookshelf full of books , a desk with a computer monitor and keyboard .
a bed with pillows ; a nightstand ; a window with curtains ! is this a door or a wall ? the floor's carpet isn't blue .
we've seen 3 chairs , 12 tables and 456 other objects . refrigerator , television , whiteboard , shower curtain , toilet , bathtub .
café crème , naïve résumé , über . an open-vocabulary 3d semantic map of the scene .""".lower()

PHRASES = ["a photo of a chair", "A Photo of the TABLE.", "there is a sofa in the scene", "shower curtain", "the floor's carpet isn't blue!",
           "3 chairs, 12 tables and 456 objects", "  spaced    out \t text\n", "refrigerator", "café crème", "an open-vocabulary 3d map",
           "xyzzy qwertyuiop", "we've", "", "it's a nightstand; a window?!", "television_set other_furniture", "&amp; books &lt;3"]


def _train_bpe(corpus, n_merges):
    from ovo_amd.encoders.tokenizer import bytes_to_unicode
    import regex
    enc = bytes_to_unicode()
    pat = regex.compile(r"'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+")
    words = collections.Counter()
    for w in pat.findall(corpus):
        m = [enc[b] for b in w.encode("utf-8")]
        words[tuple(m[:-1] + [m[-1] + "</w>"])] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        words = new
    return merges


@pytest.fixture(scope="module")
def merges():
    return _train_bpe(CORPUS, 200)


def test_byte_table_is_a_bijection():
    from ovo_amd.encoders.tokenizer import bytes_to_unicode
    t = bytes_to_unicode()
    assert len(t) == 256 and len(set(t.values())) == 256 and t[ord("a")] == "a" and t[ord(" ")] == chr(256 + 32)


def test_clip_bpe_vs_huggingface(merges):
    from transformers import CLIPTokenizer
    from ovo_amd.encoders.tokenizer import SimpleTokenizer
    ctx = 24
    tok = SimpleTokenizer(merges, context_length=ctx)
    vocab = {t: i for t, i in tok.encoder.items()}
    sot, eot = vocab.pop("<start_of_text>"), vocab.pop("<end_of_text>")
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = sot, eot
    hf = CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges])
    assert tok.eot == tok.vocab_size - 1 and tok.sot == tok.vocab_size - 2
    for p in PHRASES:
        ref = hf(p)["input_ids"]
        clean = p.replace("&amp;", "&").replace("&lt;", "<")           # HF does not unescape HTML entities; open_clip's cleaner does
        ref = hf(clean)["input_ids"] if clean != p else ref
        got = [tok.sot] + tok.encode(p) + [tok.eot]
        assert got == ref, (p, got, ref)
    ids = tok(PHRASES)
    assert ids.shape == (len(PHRASES), ctx) and ids.dtype == torch.long
    for r, p in enumerate(PHRASES):
        n = min(len(tok.encode(p)) + 2, ctx)
        assert ids[r, 0] == tok.sot and ids[r, n - 1] == tok.eot and int(ids[r].argmax()) == n - 1 and torch.all(ids[r, n:] == 0)
    long = tok(["chair " * 100])                                         # truncation keeps end-of-text in the last slot
    assert long.shape == (1, ctx) and long[0, -1] == tok.eot and long[0, 0] == tok.sot
    assert tok.decode(tok.encode("a photo of a chair")).strip() == "a photo of a chair"
    assert torch.equal(tok("a lamp"), tok(["a lamp"]))                   # the reference calls it once per phrase (clip_generator.py:170)


def test_clip_bpe_file_format(tmp_path, merges):
    """The vocabulary arrives as open_clip's bpe_simple_vocab file: header line, `left right` per line, optionally gzipped;
    only the first vocab_size - 258 merges are used."""
    import gzip
    from ovo_amd.encoders.tokenizer import SimpleTokenizer, read_merges
    text = "#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n"
    plain, gz = tmp_path / "bpe.txt", tmp_path / "bpe.txt.gz"
    plain.write_text(text, encoding="utf-8")
    with gzip.open(gz, "wb") as f:
        f.write(text.encode("utf-8"))
    v = 512 + 2 + len(merges)
    assert read_merges(str(plain), v) == read_merges(str(gz), v) == [tuple(m) for m in merges]
    assert len(read_merges(str(gz), v - 50)) == len(merges) - 50
    a, b = SimpleTokenizer(str(gz), 16, v), SimpleTokenizer(merges, 16)
    assert torch.equal(a(PHRASES), b(PHRASES)) and a.vocab_size == v


def test_siglip_sentencepiece_vs_huggingface(tmp_path):
    import sentencepiece as spm
    from transformers import SiglipTokenizer
    from ovo_amd.encoders.tokenizer import SigLIPTokenizer, canonicalize_text
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("\n".join(canonicalize_text(l) for l in CORPUS.split(".") if l.strip()) * 4, encoding="utf-8")
    prefix = str(tmp_path / "toy")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=prefix, vocab_size=120, model_type="unigram", pad_id=0, eos_id=1, unk_id=2,
                                   bos_id=-1, hard_vocab_limit=False, minloglevel=2)
    ctx = 16
    tok = SigLIPTokenizer(prefix + ".model", context_length=ctx)
    hf = SiglipTokenizer(prefix + ".model", model_max_length=ctx)
    assert tok.eos == 1 == hf.eos_token_id
    assert canonicalize_text("A photo_of the TABLE!!  ,ok") == "a photo of the table ok"
    for p in PHRASES:
        if not p.strip() or "&" in p or "_" in p:                # HF drops "_"; open_clip's cleaner (the reference's) turns it into a space
            continue
        ref = hf(p, padding="max_length", max_length=ctx, truncation=True)["input_ids"]
        got = tok(p)[0].tolist()
        assert got == ref, (p, got, ref)
    ids = tok(PHRASES)
    assert ids.shape == (len(PHRASES), ctx) and torch.all(ids[12] == 1)      # the empty phrase is all </s>


def test_get_tokenizer_contexts(tmp_path, merges):
    from ovo_amd.encoders.tokenizer import get_tokenizer
    text = "#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n"
    p = tmp_path / "bpe.txt"
    p.write_text(text, encoding="utf-8")
    assert get_tokenizer("ViT-B-16-qg", str(p)).context_length == 77
    assert get_tokenizer("PE-Core-L-14-336", str(p)).context_length == 32
    assert get_tokenizer("ViT-H-14-378qg", str(p))("a chair").shape == (1, 77)
