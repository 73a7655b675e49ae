"""ctypes binding of libovo_hip.so (the C ABI declared in include/ovo_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.  Tensors cross
the boundary as raw device pointers (`tensor.data_ptr()`) plus sizes and torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libovo_hip.so")
ABI_VERSION = 12
E_UNSUPPORTED = -3          # OVO_E_UNSUPPORTED: the entry point does not cover this shape; the caller takes its general path


class OvoHipError(RuntimeError):
    pass


class Camera(C.Structure):
    """ovo_camera_t"""
    _fields_ = [("aabb", C.c_float * 6), ("planes", C.c_float * 24), ("w2c", C.c_float * 16),
                ("K", C.c_float * 9), ("th", C.c_float), ("h", C.c_int32), ("w", C.c_int32)]


class Ratio(C.Structure):
    """ovo_ratio_t"""
    _fields_ = [("enabled", C.c_int32), ("r_h", C.c_float), ("r_w", C.c_float), ("crop_edge", C.c_int32)]


_P, _I64, _I32, _F32, _SZ = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t
_CAM = C.POINTER(Camera)


class Gemm(C.Structure):
    """ovo_gemm_t"""
    _fields_ = [("A", _P), ("lda", _I64), ("W", _P), ("ldw", _I64), ("bias", _P), ("C", _P), ("ldc", _I64),
                ("add", _P), ("ld_add", _I64), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("act", C.c_int32), ("alpha", _F32)]


class Rope(C.Structure):
    """ovo_rope_t"""
    _fields_ = [("cos", _P), ("sin", _P), ("T", C.c_int32), ("hd", C.c_int32), ("cols", C.c_int32), ("t0", C.c_int32)]


class Window(C.Structure):
    """ovo_window_t"""
    _fields_ = [(n, C.c_int32) for n in ("B", "H", "W", "wh", "ww")]


class Attention(C.Structure):
    """ovo_attention_t"""
    _fields_ = [("q", _P), ("k", _P), ("v", _P), ("o", _P)] + \
               [(n, _I64) for n in ("q_sb", "q_sh", "q_st", "k_sb", "k_sh", "k_st", "v_sb", "v_sh", "v_st", "o_sb", "o_sh", "o_st")] + \
               [(n, C.c_int32) for n in ("B", "H", "Tq", "Tk", "hd")] + [("scale", _F32), ("causal", C.c_int32)]


class VitConfig(C.Structure):
    """ovo_vit_config_t"""
    _fields_ = [(n, C.c_int32) for n in ("image_size", "patch", "width", "layers", "heads", "mlp_dim", "out_dim", "n_prefix",
                                          "act", "pre_ln", "use_rope", "pool", "kpad")] + [("ln_eps", _F32), ("q_prescaled", C.c_int32)]


class VitLayer(C.Structure):
    """ovo_vit_layer_t"""
    _fields_ = [(n, _P) for n in ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_g", "ln2_b",
                                   "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                   "qkv_wf", "qkv_bf", "qkv_cs", "fc1_wf", "fc1_bf", "fc1_cs")]      # ABI 12: the LayerNorm fold's weights (or NULL)


class VitWeights(C.Structure):
    """ovo_vit_weights_t"""
    _fields_ = [(n, _P) for n in ("patch_w", "patch_b", "prefix", "pos", "ln_pre_g", "ln_pre_b", "ln_post_g", "ln_post_b",
                                   "proj_w", "rope_cos", "rope_sin")] + [("layers", C.POINTER(VitLayer))] + \
        [(n, _P) for n in ("map_q", "map_kv_w", "map_kv_b", "map_proj_w", "map_proj_b", "map_ln_g", "map_ln_b",
                           "map_fc1_w", "map_fc1_b", "map_fc2_w", "map_fc2_b")]


class HieraConfig(C.Structure):
    """ovo_hiera_config_t"""
    _fields_ = [("image_size", C.c_int32), ("dims", C.c_int32 * 4), ("heads", C.c_int32 * 4), ("blocks", C.c_int32 * 4),
                ("window", C.c_int32 * 4), ("n_global", C.c_int32), ("global_blocks", C.c_int32 * 8), ("fpn_dim", C.c_int32),
                ("hi_res", C.c_int32), ("ln_eps", _F32), ("q_prescaled", C.c_int32)]


class HieraBlock(C.Structure):
    """ovo_hiera_block_t"""
    _fields_ = [(n, _P) for n in ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_g", "ln2_b",
                                   "fc1_w", "fc1_b", "fc2_w", "fc2_b", "res_w", "res_b")]


class HieraWeights(C.Structure):
    """ovo_hiera_weights_t"""
    _fields_ = [("patch_w", _P), ("patch_b", _P), ("pos", _P), ("blocks", C.POINTER(HieraBlock)),
                ("neck_w", _P * 4), ("neck_b", _P * 4), ("s0_w", _P), ("s0_b", _P), ("s1_w", _P), ("s1_b", _P)]


class MapRef(C.Structure):
    """ovo_map_ref_t"""
    _fields_ = [("xyz", _P), ("ids", _P), ("ins", _P), ("rgb", _P), ("cap", _I64), ("state", _P), ("n", _I64), ("next_id", _I64)]


class MapStep(C.Structure):
    """ovo_map_step_t"""
    _fields_ = [("map", MapRef), ("depth", _P), ("rgb", _P), ("h", C.c_int32), ("w", C.c_int32), ("cam", Camera),
                ("K", C.c_float * 9), ("c2w", C.c_float * 16), ("ds", C.c_int32), ("erode", C.c_int32), ("n_upper", _I64),
                ("explained", _P), ("ws", _P), ("ws_bytes", _SZ), ("result_host", _P), ("seq", _I64)]


class TrackStep(C.Structure):
    """ovo_track_step_t"""
    _fields_ = [("map", MapRef), ("depth", _P), ("filter_depth", C.c_int32), ("depth_scratch", _P), ("cam", Camera), ("ratio", Ratio),
                ("seg_map", _P), ("seg_h", C.c_int32), ("seg_w", C.c_int32), ("masks", _P), ("n_masks", C.c_int32), ("pixels", _I64),
                ("point_seg", _P), ("ws", _P), ("ws_bytes", _SZ), ("hist_cols", C.c_int32), ("track_th", C.c_int32),
                ("next_ins", _P), ("next_ins_host", C.c_int32), ("n_upper", _I64), ("result_host", _P), ("seq", C.c_int32),
                ("hits", _P), ("n_hits", _P), ("hit_shard_rank", C.c_int32), ("hit_shard_count", C.c_int32), ("hit_shard_block", C.c_int32)]


class RoundChain(C.Structure):
    """ovo_round_chain_t"""
    _fields_ = [("params_host", _P), ("barrier", _P), ("arrivals", C.c_uint64), ("next_slot", C.c_uint32), ("workgroups", C.c_int32)]


DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.uint8: 3}

_SIGNATURES = {
    "ovo_hip_last_error": (C.c_char_p, []),
    "ovo_hip_abi_version": (_I32, []),
    "ovo_marker": (_I32, [_I32, _P]),
    "ovo_profile_start": (_I32, []),
    "ovo_profile_stop": (_I32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), _I32]),
    "ovo_profile_bytes": (_I32, [C.POINTER(C.c_double), _I32]),
    "ovo_compact_workspace_bytes": (_SZ, [_I64]),
    "ovo_frustum_ids": (_I32, [_P, _I64, _CAM, _P, _P, _P, _SZ, _P]),
    "ovo_project_points": (_I32, [_P, _I64, _I32, _CAM, _P, _P]),
    "ovo_match_points": (_I32, [_P, _I64, _I32, _CAM, _P, _P, _P, _P, _P, _SZ, _P]),
    "ovo_depth_filter": (_I32, [_P, _I32, _I32, _I32, _F32, _F32, _P, _P]),
    "ovo_track_project": (_I32, [_P, _P, _I64, _CAM, _P, _P, _I32, _I32, Ratio, _P, _P, _I32, _I32, _P, _P]),
    "ovo_vote_stats": (_I32, [_P, _I32, _I32, _P, _I64, _P, _P]),
    "ovo_assign_instances": (_I32, [_P, _P, _I64, _P, _I32, _P, _P, _P]),
    "ovo_map_explained": (_I32, [_P, _I64, _CAM, _P, _P, _P]),
    "ovo_map_backproject": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ovo_map_step": (_I32, [C.POINTER(MapStep), _P]),
    "ovo_track_step": (_I32, [C.POINTER(TrackStep), _P]),
    "ovo_track_workspace_bytes": (_SZ, [_I32, _I32]),
    "ovo_keyframe_step": (_I32, [C.POINTER(MapStep), C.POINTER(TrackStep), _P]),
    "ovo_round_chain_params_bytes": (_SZ, []),
    "ovo_round_chain": (_I32, [C.POINTER(RoundChain), C.POINTER(MapStep), C.POINTER(TrackStep), _I32, _P]),
    "ovo_host_alloc": (_P, [_SZ]),
    "ovo_host_free": (None, [_P]),
    "ovo_host_wait32": (_I32, [_P, C.c_int32, _I64]),
    "ovo_host_wait64": (_I32, [_P, _I64, _I64]),
    "ovo_fuse_views": (_I32, [_P, _I32, _P, _P, _I32, _I32, _P, _P, _P, _P]),
    "ovo_fuse_views_add": (_I32, [_P, _I32, _P, _P, _P, _I32, _P, _P, _P, _P]),
    "ovo_scatter_accum": (_I32, [_P, _I64, _P, _I32, _P, _I32, _P, _P, _P]),
    "ovo_similarity": (_I32, [_P, _I32, _I64, _I32, _P, _I32, _P, _I32, _F32, _F32, _F32, _P, _P, _P, _P]),
    "ovo_similarity_rows": (_I32, [_P, _I32, _P, _P, _I64, _I32, _P, _I32, _P, _I32, _F32, _F32, _F32, _P, _P, _P]),
    "ovo_scatter_accum_touched": (_I32, [_P, _I64, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "ovo_scatter_accum_query": (_I32, [_P, _P, _I64, _P, _P, _I32, _P, _I32, _P, _P, _I32, _I32, _I32, _P, _I32, _I32, _F32, _F32, _F32, _P, _P, _P]),
    "ovo_mask_boxes": (_I32, [_P, _I32, _I32, _I32, _P, _P]),
    "ovo_mask_crops": (_I32, [_P, _I32, _I32, _I32, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "ovo_row_argmax": (_I32, [_P, _I64, _I32, _I32, _F32, _F32, _F32, _P, _P, _P]),
    "ovo_mask_intersections": (_I32, [_P, _I32, _I64, _P, _P]),
    "ovo_pack_masks": (_I32, [_P, _I32, _I64, _P, _I64, _P]),
    "ovo_unpack_masks": (_I32, [_P, _I32, _I64, _I64, _P, _P]),
    "ovo_mask_or": (_I32, [_P, _I64, _P, _I32, _P]),
    "ovo_gather_rows": (_I32, [_P, _I64, _P, _I32, _P, _P]),
    "ovo_mask_area": (_I32, [_P, _I64, _P, _I32, _P, _P]),
    "ovo_gemm": (_I32, [C.POINTER(Gemm), _P]),
    "ovo_gemm_argmax": (_I32, [C.POINTER(Gemm), _P, _I32, _I32, _P]),
    "ovo_gemm_rope": (_I32, [C.POINTER(Gemm), C.POINTER(Rope), _P]),
    "ovo_gemm_periodic": (_I32, [C.POINTER(Gemm), _I64, _P]),
    "ovo_gemm_unwindow": (_I32, [C.POINTER(Gemm), C.POINTER(Window), _P]),
    "ovo_gemm_fold_out": (_I32, [C.POINTER(Gemm), _P, _I64, _P, _I64, _P]),
    "ovo_gemm_fold_stats": (_I32, [_P, _I64, _I32, _I32, _P, _I64, _P, _P]),
    "ovo_gemm_fold_in": (_I32, [C.POINTER(Gemm), C.POINTER(Rope), _P, _I64, _I32, _I32, _P, _F32, _P]),
    "ovo_gemm_rowln": (_I32, [C.POINTER(Gemm), C.POINTER(Window), _P, _P, _F32, _P, _I64, _P]),
    "ovo_gemm_f32a": (_I32, [C.POINTER(Gemm), C.POINTER(Window), _P, C.c_int, _P, _P, C.c_float, C.c_int, C.c_int, _P]),
    "ovo_decode_best": (_I32, [_P, _I64, _F32, _P, _P, _P]),
    "ovo_attention": (_I32, [C.POINTER(Attention), _P]),
    "ovo_layernorm": (_I32, [_P, _I64, _I64, _I32, _P, _P, _F32, _P, _I64, _I32, _P]),
    "ovo_vit_embed": (_I32, [_P, _P, _I32, _P, _I32, _I32, _I32, _P, _P, _F32, _P, _P]),
    "ovo_im2col": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "ovo_resize_normalize": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _F32,
                                    C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "ovo_resize_normalize_batch": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _I32, _I32, _F32, _P, _P, _P]),
    "ovo_resize_window_normalize": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F32,
                                           C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "ovo_rope_qk": (_I32, [_P, _I32, _I32, _I32, _I32, _P, _P, _I32, _P]),
    "ovo_feature_masks": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _P]),
    "ovo_stitch_tokens_t": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "ovo_unit_tokens": (_I32, [_P, _I32, _I32, _I32, _P, _P, _P]),
    "ovo_global_patch_filter": (_I32, [_P, _P, _I32, _I32, _I32, _F32, _P, _P]),
    "ovo_scale_rows_bf16": (_I32, [_P, _P, _I32, _I32, _P, _P]),
    "ovo_mlp_f32": (_I32, [_P, _I64, _I32, _P, _P, _F32, _P, _I64, _P, _I32, _P, _I64, _P, _P]),
    "ovo_window_attention_f32": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _F32, _P, _I64, _P, _P, _I32, _P]),
    "ovo_l2_normalize_rows": (_I32, [_P, _I64, _I32, _P, _P]),
    "ovo_cast_f32": (_I32, [_P, _I64, _P, _I32, _P]),
    "ovo_row_epilogue": (_I32, [_P, _I64, _I32, _P, _I64, _P, _P, _F32, _P, _I64, _P, _P, _P, _P]),
    "ovo_sam_upscale_ln": (_I32, [_P, _P, _P, _P, _P, _F32, _I64, _I32, _I32, _P, _P]),
    "ovo_sam_upscale_masks": (_I32, [_P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _P, _P]),
    "ovo_sam_linear": (_I32, [_P, _P, _P, _P, _I64, _I32, _P, _I32, _I64, _I32, _I32, _P]),
    "ovo_sam_proj_ln": (_I32, [_P, _P, _P, _P, _P, _I64, _P, _P, _F32, _P, _I64, _P, _P, _P, _I64, _I32, _I32, _P]),
    "ovo_sam_up1_ln": (_I32, [_P, _P, _P, _P, _P, _P, _F32, _I64, _I32, _I32, _I32, _P, _P]),
    "ovo_sam_up2_masks": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "ovo_instance_moments": (_I32, [_P, _P, _I64, _I32, _P, _P, _P]),
    "ovo_near_fraction": (_I32, [_P, _P, _P, _I32, _I64, _F32, _P, _P]),
    "ovo_remap_instances": (_I32, [_P, _I64, _P, _I32, _P]),
    "ovo_sam_i2t_attention": (_I32, [_P, _I64, _I32, _P, _P, _P, _I64, _I32, _I32, _I32, _F32, _P]),
    "ovo_sam_t2i_attention": (_I32, [_P, _P, _P, _I64, _I32, _P, _I64, _I32, _I32, _I32, _F32, _P]),
    "ovo_paint_segmap": (_I32, [_P, _I32, _I64, _P, _P]),
    "ovo_amg_mask_stats": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _F32, _F32, _P, _P]),
    "ovo_amg_binarize": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _F32, _P, _P]),
    "ovo_vit_workspace_bytes": (_SZ, [C.POINTER(VitConfig), _I32]),
    "ovo_vit_forward": (_I32, [C.POINTER(VitConfig), C.POINTER(VitWeights), _P, _I32, _P, _P, _SZ, _P]),
    "ovo_hiera_workspace_bytes": (_SZ, [C.POINTER(HieraConfig), _I32]),
    "ovo_hiera_forward": (_I32, [C.POINTER(HieraConfig), C.POINTER(HieraWeights), _P, _I32, _P, _P, _P, _P, _SZ, _P]),
    "ovo_hiera_patch_embed": (_I32, [_P, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load() -> C.CDLL:
    """Load the library (no compute, works without a GPU).  Raises OvoHipError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise OvoHipError(f"{SO_PATH} is missing: run `python -m ovo_amd.build` (needs hipcc). "
                          "ovo_amd has no CPU fallback.")
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype, fn.argtypes = res, args
    if lib.ovo_hip_abi_version() != ABI_VERSION:
        raise OvoHipError(f"ABI mismatch: library {lib.ovo_hip_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        raise OvoHipError(f"libovo_hip error {status}: {load().ovo_hip_last_error().decode()}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> C.c_void_p:
    """hipStream_t of torch's current stream (the raw handle: ~1 us instead of ~9 us for a torch.cuda.Stream object, 20+ times per keyframe)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def dev(t: torch.Tensor, dtype: torch.dtype, name: str = "tensor") -> torch.Tensor:
    """Validate a tensor that is about to cross the C ABI."""
    if not t.is_cuda:
        raise OvoHipError(f"{name} must live on the GPU (got {t.device}); ovo_amd has no CPU path")
    if t.dtype != dtype:
        raise OvoHipError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise OvoHipError(f"{name} must be contiguous")
    return t


def gather_rows(src: torch.Tensor, rows) -> torch.Tensor:
    """src[rows] along dim 0 through `ovo_gather_rows` (torch's index_select picks size-dependent kernel variants whose
    first use lazily loads a code object: a 100+ ms stall in the middle of a sequence on ROCm)."""
    n = len(rows)
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if n == 0:
        return out
    row_bytes = src[0].numel() * src.element_size()
    idx = torch.tensor(list(rows), dtype=torch.int32).to(src.device, non_blocking=True)
    check(load().ovo_gather_rows(ptr(src), row_bytes, ptr(idx), n, ptr(out), stream()))
    return out


_ws_cache = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only per-device scratch buffer (u8)."""
    key = str(device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class PinnedRing:
    """Result blocks of the keyframe chain (`ovo_map_step` / `ovo_track_step`): a ring of fixed-size slots in pinned, device-visible host
    memory (`ovo_host_alloc`).  The last workgroup of a chain writes a slot and then its sequence word; `wait` spins on that word
    (no runtime call, GIL released) and returns the slot as a numpy view."""

    def __init__(self, slot_items: int, np_dtype, slots: int = 64):
        import numpy as np
        self.np = np
        self.dtype = np.dtype(np_dtype)
        self.slot_items, self.slots = int(slot_items), int(slots)
        nbytes = self.slot_items * self.slots * self.dtype.itemsize
        self.base = load().ovo_host_alloc(nbytes)
        if not self.base:
            raise OvoHipError(load().ovo_hip_last_error().decode())
        ctype = C.c_int64 if self.dtype.itemsize == 8 else C.c_int32
        self.view = np.ctypeslib.as_array(C.cast(self.base, C.POINTER(ctype)), shape=(self.slots, self.slot_items))
        self._wait = load().ovo_host_wait64 if self.dtype.itemsize == 8 else load().ovo_host_wait32
        self.seq = 0

    def next(self):
        """(sequence number, slot pointer) of the next call; the slot's sequence word is cleared first."""
        self.seq += 1
        k = self.seq % self.slots
        self.view[k, 0] = 0
        return self.seq, self.base + k * self.slot_items * self.dtype.itemsize

    def done(self, seq: int) -> bool:
        return int(self.view[seq % self.slots, 0]) == seq

    def wait(self, seq: int, timeout_us: int = 20_000_000):
        k = seq % self.slots
        check(self._wait(self.base + k * self.slot_items * self.dtype.itemsize, seq, timeout_us))
        return self.view[k]

    def __del__(self):
        try:
            if getattr(self, "base", None):
                load().ovo_host_free(self.base)
        except Exception:
            pass


LOG2E = 1.4426950408889634


def q_prescale_enabled() -> bool:
    """The encoders fold log2(e) / sqrt(head_dim) into the q rows of their QKV projections at load time (OVO_Q_PRESCALE=0: the
    attention kernel multiplies the bf16 queries itself -- a second rounding of every query; kept for A/B measurements)."""
    return os.environ.get("OVO_Q_PRESCALE", "1") != "0"


def fold_layernorm(weight: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """LN(x) . W^T + b = rstd (x . W'^T - mean colsum(W')) + b' (ovo_vit_layer_t.qkv_wf ..., ABI 12): returns W' = W . gamma (f32; the caller rounds it to bf16
    ONCE), b' = b + W . beta (f32, taken in f64) and the row sums of the ROUNDED W' (f32, taken in f64) -- the product multiplies the rounded matrix, so the
    mean's coefficient has to be its sum, not the sum of the unrounded one."""
    w = weight.detach().double()
    wf = (w * gamma.detach().double()[None, :]).float()
    bf = (bias.detach().double() + w @ beta.detach().double()).float()
    cs = wf.to(torch.bfloat16).double().sum(dim=1).float()
    return wf, bf, cs


def fold_q_scale(weight: torch.Tensor, bias: Optional[torch.Tensor], q_rows: int, head_dim: int):
    """(weight, bias) f32 copies whose first `q_rows` output rows carry log2(e) / sqrt(head_dim): softmax(q k^T / sqrt(hd)) =
    softmax2(q' k^T) with q' = q log2(e) / sqrt(hd), so ovo_attention (scale = 0) applies no factor to its bf16 queries.  The
    product is taken in f32 BEFORE the matrix is rounded to bf16 -- one rounding per weight, as without the fold."""
    c = LOG2E / float(head_dim) ** 0.5
    w = weight.detach().float().clone()
    w[:q_rows] *= c
    b = None
    if bias is not None:
        b = bias.detach().float().clone()
        b[:q_rows] *= c
    return w, b
