"""Per-mask open-vocabulary descriptors and text similarity on MI355X.

Mirror of the reference's `ovo/entities/clip_generator.py:CLIPGenerator` (same constructor, attributes and
methods).  The image tower is `ovo_amd.encoders.vit.HipViT`; region descriptors come from
`PETextRegion` (embed_type: TextRegion, the configured default, ovo.yaml:45) or from masked / bbox crops
(the other embed types, clip_generator.py:136-158).

Offline limits, stated rather than hidden:
  * no checkpoints can be downloaded -> weights are loaded from `config["weights_path"]` (an open_clip-style
    `visual.*` state dict) when given, else seeded random weights of the same architecture;
  * the text tower + BPE tokenizer are a "next" row: `text_encoder` may be injected
    (callable: list[str] -> [n, D] tensor); the default is a deterministic hash embedding so that the
    query path (template ensembling, normalisation, similarity, argmax) runs end to end.
"""
from __future__ import annotations

import hashlib
import os
from typing import Callable, Dict, List, Optional, Union

import torch

from .. import _lib as L
from ..encoders.vit import SPECS, HipViT
from ..utils import clip_utils, segment_utils
from .textregion import PETextRegion

_CARD_ALIASES = {"PE-Core-L-14-336": "PE-Core-L14-336"}


def hash_text_encoder(dim: int) -> Callable[[List[str]], torch.Tensor]:
    """Deterministic stand-in for `model.encode_text`: a seeded Gaussian vector per distinct string."""
    def encode(texts: List[str]) -> torch.Tensor:
        rows = []
        for s in texts:
            seed = int.from_bytes(hashlib.sha256(s.encode()).digest()[:8], "little") % (2 ** 63)
            rows.append(torch.randn(dim, generator=torch.Generator().manual_seed(seed)))
        return torch.stack(rows)
    return encode


class CLIPGenerator:
    def __init__(self, config: Dict, device: str = "cuda", encoder: Optional[HipViT] = None,
                 text_encoder: Optional[Callable[[List[str]], torch.Tensor]] = None):
        self.config = config
        self.device = device
        self.embed_type = config.get("embed_type", "vanilla")
        self.mask_res = config.get("mask_res", 384)
        if self.embed_type == "learned":
            raise NotImplementedError("embed_type 'learned' needs the weights-predictor checkpoint (data/input/ReadMe.md:10), "
                                      "which is not available offline; out of scope (SURVEY.md §2 row 17)")
        self.w_masked = config.get("w_masked", 0.4418)
        self.w_global = config.get("w_global", 0.1)
        self.model_card = config.get("model_card", "SigLIP-384")
        card = _CARD_ALIASES.get(self.model_card, self.model_card)
        if encoder is None:
            if card not in SPECS:
                raise NotImplementedError(f"model card {self.model_card}: supported here: {sorted(SPECS)}")
            state = None
            path = config.get("weights_path")
            if path and os.path.exists(path):
                state = torch.load(path, map_location="cpu")
                state = {k[len("visual."):] if k.startswith("visual.") else k: v for k, v in state.items()}
            encoder = HipViT(SPECS[card], state, device=device, seed=config.get("seed", 0))
        self.model = encoder
        self.clip_dim = 1024 if self.embed_type == "TextRegion" else encoder.spec.out_dim
        if self.embed_type == "TextRegion":
            self.textregion = PETextRegion(encoder, model_card="PE" if not card.startswith("PE") else card,
                                           resize_method=config.get("resize_method", "multi_resolution"),
                                           remove_global_patch=config.get("remove_global_patch", False),
                                           project_and_normalize=config.get("project_and_normalize", True),
                                           share_identical_crops=config.get("share_identical_crops"))
            self.clip_dim = self.textregion.out_dim
        if text_encoder is None and config.get("vocab_path"):
            # the real text side (clip_generator.py:161-173): tokenizer file + text tower of the same card
            from ..encoders.text import SPECS as TEXT_SPECS, HipTextEncoder
            from ..encoders.tokenizer import get_tokenizer
            tcard = card if card in TEXT_SPECS else {"ViT-H-14-qg": "ViT-H-14", "ViT-H-14-378qg": "ViT-H-14"}.get(card, card)
            if tcard not in TEXT_SPECS:
                raise NotImplementedError(f"no text tower for model card {self.model_card}: supported here: {sorted(TEXT_SPECS)}")
            tstate = None
            tpath = config.get("text_weights_path") or config.get("weights_path")
            if tpath and os.path.exists(tpath):
                tstate = {k: v for k, v in torch.load(tpath, map_location="cpu").items() if not k.startswith("visual.")}
            text_encoder = HipTextEncoder(TEXT_SPECS[tcard], tstate, device=device, seed=config.get("seed", 0),
                                          tokenizer=get_tokenizer(self.model_card, config["vocab_path"]))
        self._encode_text = text_encoder or hash_text_encoder(self.clip_dim)
        self.tokenizer = None
        if self.model_card.startswith("SigLIP"):
            self.get_similarity = clip_utils.siglip_cosine_similarity
            self.similarity_args = (float(config.get("logit_scale") or 4.6052), float(config.get("logit_bias") or -10.0))
        else:
            self.get_similarity = clip_utils.clip_cosine_similarity
            self.similarity_args = ()

    @property
    def get_clip_dim(self) -> int:
        return self.clip_dim

    # ------------------------------------------------------------------ device moves (clip_generator.py:78-109)
    def to(self, device: str) -> None:
        return self.cuda() if "cuda" in device else self.cpu()

    def cpu(self) -> None:
        self.device = "cpu"           # weights stay resident on the GPU: there is no CPU execution path

    def cuda(self) -> None:
        self.device = "cuda"

    # ------------------------------------------------------------------ image side
    @torch.no_grad()
    def encode_image(self, input: torch.Tensor) -> torch.Tensor:
        """Reference: clip_generator.py:112-122.  [B, 3, h, w] (or [3, h, w]) in [0, 1] -> [B, D]: the card's kept open_clip transforms
        (Resize on the shorter side / squash, antialiased bicubic / bilinear, CenterCrop, Normalize; clip_utils.py:83-84), then the tower."""
        if input.dim() == 3:
            input = input[None]
        x = L.dev(input.float().contiguous(), torch.float32, "input")
        return self.model.forward(self.model.preprocess_clip(x))

    @torch.no_grad()
    def extract_clip(self, image: torch.Tensor, binary_maps: torch.Tensor, return_all: bool = False) -> torch.Tensor:
        """Reference: clip_generator.py:125-158.  image [3, H, W] in 0..255 (u8 or float), binary_maps bool [N, H, W]
        -> f32 [N, clip_dim] on the GPU."""
        if binary_maps.shape[0] == 0:
            return torch.zeros((0, self.clip_dim), dtype=torch.float32, device=binary_maps.device)
        if self.embed_type == "TextRegion":
            img = image if image.dtype == torch.uint8 else image.float()
            return self.textregion.predict(img.contiguous(), binary_maps, scale=1.0 / 255.0)
        from ..utils import segment_utils
        also_bbox = self.embed_type != "vanilla"
        seg = segment_utils.segmap2segimg(binary_maps, image, also_bbox, out_l=self.mask_res) / 255.0
        img = image.float() if image.dtype != torch.float32 else image
        norm = torch.nn.functional.normalize
        if not also_bbox:
            return norm(self.encode_image(seg[:, :3]), p=2, dim=-1)
        n = seg.shape[0]
        clip_g = norm(self.encode_image(img[None] / 255.0), p=2, dim=-1)
        both = norm(self.encode_image(torch.cat([seg[:, :3], seg[:, 3:]], dim=0)), p=2, dim=-1)
        if return_all:
            return torch.cat([clip_g.repeat(n, 1)[:, None], both[:n][:, None], both[n:][:, None]], dim=1)
        return clip_utils.fuse_clips(clip_g.repeat(n, 1), both[:n], both[n:], self.embed_type, self.w_masked, self.w_global)

    # ------------------------------------------------------------------ text side
    @torch.no_grad()
    def get_txt_embedding(self, text_list: List[str]) -> torch.Tensor:
        """Reference: clip_generator.py:161-173 -> unit-norm [n, D]."""
        e = self._encode_text(list(text_list)).float()
        return e / e.norm(dim=-1, keepdim=True)

    def embed_queries(self, txt_queries: List[str], templates: Union[str, List[str]] = ("{}",)) -> torch.Tensor:
        """Per query: mean over templates of the unit-norm embeddings, re-normalised (clip_generator.py:189-196)."""
        if isinstance(templates, str):
            templates = [templates]
        rows = [torch.nn.functional.normalize(self.get_txt_embedding([t.format(q) for t in templates]).mean(0, keepdim=True), p=2, dim=-1)
                for q in txt_queries]
        return torch.cat(rows, dim=0)

    @torch.no_grad()
    def get_embed_txt_similarity(self, ins_descriptors: torch.Tensor, txt_queries: List[str],
                                 templates: Union[str, List[str]] = ['{}']) -> torch.Tensor:
        """Reference: clip_generator.py:176-199 -> [N, Q] similarity map."""
        txt = self.embed_queries(txt_queries, templates).to(ins_descriptors.device)
        return self.get_similarity(txt, ins_descriptors, *self.similarity_args)

    @torch.no_grad()
    def classify(self, ins_descriptors: torch.Tensor, classes: List[str], templates="This is a photo of a {}", th: float = 0.0):
        """Similarity + row argmax + threshold in one kernel launch (ovo.py:486-491)."""
        txt = self.embed_queries(classes, templates).to(ins_descriptors.device)
        sig = len(self.similarity_args) == 2
        _, cls, conf = clip_utils.similarity(ins_descriptors, txt, siglip=sig,
                                             logit_scale=self.similarity_args[0] if sig else 0.0,
                                             logit_bias=self.similarity_args[1] if sig else 0.0,
                                             want_sim=False, want_argmax=True, th=th)
        return cls, conf
