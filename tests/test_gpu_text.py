"""-m gpu: CLIP text tower (f2) on MI355X vs the fp32 oracle (pinned to HuggingFace's CLIPTextModelWithProjection)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_text_tower_hf_weights(act):
    """The HuggingFace golden weights through the HIP tower: bf16 GEMM operands, fp32 accumulation / residuals / LayerNorm."""
    from oracle import text as OT
    from ovo_amd.encoders.text import HipTextEncoder, TextSpec
    d = golden("hf_clip_text")
    pre = f"{act}:w:"
    sd = OT.hf_clip_text_to_openclip({k[len(pre):]: torch.from_numpy(d[k]) for k in d.files if k.startswith(pre)})
    spec = TextSpec("golden", 100, 16, 64, 3, int(d["heads"]), 32, act)
    enc = HipTextEncoder(spec, sd, device=DEV)
    out = enc.encode_tokens(torch.from_numpy(d[f"{act}:ids"])).cpu().numpy()
    ref = d[f"{act}:out"]
    err = np.abs(_unit(out) - _unit(ref)).max()
    print(f"{act}: max |unit embedding error| vs HuggingFace = {err:.2e}")
    assert out.shape == ref.shape and err < 1e-3 * (512 / 32) ** 0.5 * 1.5     # 1e-3 at D >= 512, scaled to this 32-d projection (as in test_gpu_encoder)


def test_siglip_text_tower_hf_weights():
    """HuggingFace SiglipTextModel golden weights through the HIP tower (bidirectional attention, last-position pooling, tanh-GELU,
    hidden 176 zero-padded to 192, projection bias)."""
    from oracle import text as OT
    from ovo_amd.encoders.text import HipTextEncoder, TextSpec
    d = golden("hf_siglip_text")
    sd = OT.hf_siglip_text_to_openclip({k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")})
    spec = TextSpec("golden", 100, 16, 64, 3, int(d["heads"]), 64, "gelu_tanh", 176, False, "last", True, 1e-6)
    out = HipTextEncoder(spec, sd, device=DEV).encode_tokens(torch.from_numpy(d["ids"])).cpu().numpy()
    err = np.abs(_unit(out) - _unit(d["out"])).max()
    print(f"max |unit embedding error| vs HuggingFace = {err:.2e}")
    assert out.shape == d["out"].shape and err < 1e-3 * (512 / 64) ** 0.5 * 1.5


@pytest.mark.parametrize("card,batch,layers", [("tiny-siglip-text", 5, None), ("SigLIP-384", 3, 2)])
def test_siglip_text_tower_vs_oracle(card, batch, layers):
    """so400m text shape: width 1152 (row kernels with 8 vectors per lane), head_dim 72, hidden 4304 -> 4320, 64 positions."""
    import dataclasses
    from oracle import text as OT
    from ovo_amd.encoders.text import SPECS, HipTextEncoder, random_state
    spec = SPECS[card]
    if layers:
        spec = dataclasses.replace(spec, layers=layers, vocab=500)
    sd = random_state(spec, seed=4)
    ids = torch.randint(2, spec.vocab, (batch, spec.context), generator=torch.Generator().manual_seed(batch))
    for r in range(batch):
        ids[r, 3 + 5 * r:] = 1                                # padded to the context length, as SigLIP's tokenizer does
    out = HipTextEncoder(spec, sd, device=DEV).encode_tokens(ids).cpu().numpy()
    ref = OT.text_forward(sd, ids, heads=spec.heads, act=spec.act, eps=spec.ln_eps, causal=False, pool="last").numpy()
    err = np.abs(_unit(out) - _unit(ref)).max()
    cos = (_unit(out) * _unit(ref)).sum(-1).min()
    print(f"{card}: max |unit embedding error| = {err:.2e}, min cosine = {cos:.6f}")
    assert err < 1e-3 * max(1.0, (512 / spec.out_dim) ** 0.5 * 1.5) and cos > 0.9999


@pytest.mark.parametrize("card,batch,t", [("tiny-text", 7, 16), ("tiny-text", 3, 9), ("ViT-B-16-qg", 4, 77)])
def test_text_tower_vs_oracle(card, batch, t):
    from oracle import text as OT
    from ovo_amd.encoders.text import SPECS, HipTextEncoder, random_state
    spec = SPECS[card]
    sd = random_state(spec, seed=3)
    g = torch.Generator().manual_seed(batch)
    ids = torch.randint(1, spec.vocab - 2, (batch, t), generator=g)
    for r in range(batch):                                # end-of-text (highest id) somewhere, padding zeros after it
        n = int(torch.randint(2, t + 1, (1,), generator=g))
        ids[r, n - 1] = spec.vocab - 1
        ids[r, n:] = 0
    enc = HipTextEncoder(spec, sd, device=DEV)
    out = enc.encode_tokens(ids).cpu().numpy()
    ref = OT.text_forward(sd, ids, heads=spec.heads, act=spec.act).numpy()
    err = np.abs(_unit(out) - _unit(ref)).max()
    cos = (_unit(out) * _unit(ref)).sum(-1).min()
    print(f"{card} B={batch} T={t}: max |unit embedding error| = {err:.2e}, min cosine = {cos:.6f}")
    assert err < 1e-3 * max(1.0, (512 / spec.out_dim) ** 0.5 * 1.5) and cos > 0.9999
    # causal: an embedding must not depend on what follows its end-of-text token
    ids2 = ids.clone()
    ids2[ids == 0] = 5
    ids2[:, 0] = ids[:, 0]
    assert torch.equal(enc.encode_tokens(ids2), enc.encode_tokens(ids)) or np.abs(enc.encode_tokens(ids2).cpu().numpy() - out).max() < 1e-6


def test_text_encoder_plugs_into_clip_generator():
    from ovo_amd.encoders.text import SPECS, HipTextEncoder
    from ovo_amd.encoders.vit import SPECS as VS, HipViT
    from ovo_amd.entities.clip_generator import CLIPGenerator
    spec = SPECS["tiny-text"]

    def tokenizer(texts):                                 # stand-in vocabulary: one id per distinct word
        vocab = {}
        rows = []
        for s in texts:
            ids = [98] + [vocab.setdefault(w, 1 + len(vocab) % 90) for w in s.split()][: spec.context - 2] + [99]
            rows.append(ids + [0] * (spec.context - len(ids)))
        return torch.tensor(rows)
    enc = HipTextEncoder(spec, None, device=DEV, tokenizer=tokenizer)
    gen = CLIPGenerator({"embed_type": "vanilla", "model_card": "tiny-clip"}, device=DEV, encoder=HipViT(VS["tiny-clip"], None, device=DEV),
                        text_encoder=enc)
    e = gen.get_txt_embedding(["a chair", "a photo of a table", "lamp"])
    assert e.shape == (3, spec.out_dim) and torch.allclose(e.norm(dim=-1), torch.ones(3, device=e.device), atol=1e-5)
    with pytest.raises(Exception):
        HipTextEncoder(spec, None, device=DEV)(["no tokenizer"])


def test_clip_generator_builds_its_text_side_from_a_vocabulary_file(tmp_path):
    """config["vocab_path"] -> tokenizer + text tower of the card; `get_embed_txt_similarity` (clip_generator.py:176-199) end to end."""
    from oracle import text as OT
    from test_tokenizer import CORPUS, _train_bpe
    from ovo_amd.encoders.text import SPECS, random_state
    from ovo_amd.encoders.vit import SPECS as VS, HipViT
    from ovo_amd.entities.clip_generator import CLIPGenerator
    merges = _train_bpe(CORPUS, 200)
    vocab = tmp_path / "bpe_simple_vocab.txt"
    vocab.write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    gen = CLIPGenerator({"embed_type": "vanilla", "model_card": "tiny-clip", "vocab_path": str(vocab), "seed": 5}, device=DEV,
                        encoder=HipViT(VS["tiny-clip"], None, device=DEV))
    phrases = ["a photo of a chair", "the table in a room", "shower curtain"]
    e = gen.get_txt_embedding(phrases).cpu().numpy()
    spec = SPECS["tiny-clip"]
    ids = gen._encode_text.tokenizer(phrases)
    assert ids.shape == (3, spec.context) and int(ids.max()) == spec.vocab - 1
    ref = OT.text_forward(random_state(spec, 5), ids, heads=spec.heads, act=spec.act).numpy()
    assert np.abs(e - _unit(ref)).max() < 1e-3 * (512 / spec.out_dim) ** 0.5 * 1.5
    desc = torch.nn.functional.normalize(torch.randn(7, spec.out_dim, generator=torch.Generator().manual_seed(1)), dim=-1).to(DEV)
    sim = gen.get_embed_txt_similarity(desc, ["chair", "table"], templates=["a photo of a {}", "there is a {} in the scene"])
    t = np.stack([_unit(_unit(OT.text_forward(random_state(spec, 5), gen._encode_text.tokenizer([f"a photo of a {q}", f"there is a {q} in the scene"]),
                                              heads=spec.heads, act=spec.act).numpy()).mean(0)) for q in ("chair", "table")])
    want = desc.cpu().numpy() @ t.T
    got = sim.cpu().numpy()
    got = got if got.shape == want.shape else got.T
    np.testing.assert_allclose(got, want, atol=5e-3)
