"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo on CPU.

The reference is single-process (no collective anywhere).  The path shards at frame granularity
(SURVEY.md §8e): the heavy per-frame work -- SAM2 encoder (+ mask generator), ViT forward, region pooling -- is independent per
frame, so in a round of N keyframes rank r owns keyframe r; the order-dependent integer passes run replicated (pipeline.py).
The ONE exchange of a round is `allgather` of the owners' descriptors (f32[128, D] per rank: KBs, latency-bound over xGMI); the dense
per-point accumulators are SHARDED by point, never reduced: every rank applies the gathered descriptors to its own rows in keyframe
order, so the shards equal a single accumulator bit for bit (a floating-point all-reduce of per-GPU partial sums would not).
`allreduce_dense_` (the bucketed sum of whole accumulators, north_star's "single RCCL reduce") stays for accumulators that were built
independently -- e.g. two maps of the same scene merged offline; it is not on the keyframe path.
No NCCL call pattern is translated from anywhere: there is none in the reference.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch
import torch.distributed as dist

FORCE_COLLECTIVES = False          # tests: issue the collective even in a one-rank group (a world-1 RCCL group runs the nccl branch on one GPU)
DENSE_BUCKET_BYTES = 256 << 20     # per all_reduce call for the dense merge (large buckets: 288 GB HBM, per-link bound)


def env_world() -> tuple:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def local_device(local_rank: int) -> int:
    """GPU index of this rank.  OVO_FORCE_DEVICE pins every rank to one GPU: the way to exercise the N > 1 code path on a
    single-GPU box (with OVO_DIST_BACKEND=gloo, since RCCL refuses two ranks on one device)."""
    forced = os.environ.get("OVO_FORCE_DEVICE")
    return int(forced) if forced is not None else local_rank


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from torchrun's environment.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("OVO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_device(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def allreduce_sum_(tensors: Sequence[torch.Tensor]) -> None:
    """In-place sum over ranks of a few small tensors, packed into ONE collective."""
    if world_size() == 1 or not tensors:
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].reshape(t.shape).to(t.dtype))
        off += n


def allgather_rows(rows: torch.Tensor) -> torch.Tensor:
    """Fixed-size per-step exchange: every rank contributes `rows` [R, C] (same shape on every rank, unused rows flagged
    by the caller) and receives all of them, [world * R, C], rank-major.  A keyframe touches a few dozen instance
    descriptors, so gathering the touched rows (R x (1 + D) floats per rank, ~0.5 MB at D = 1024) moves 30x less than
    sum-reducing the whole instance table (16 MB), and the result is the same table update on every rank."""
    if world_size() == 1:
        return rows
    # expressed as ONE sum-reduce of a zero buffer in which each rank fills its own slice: the same bytes on the wire as an
    # all-gather for a ring, and the only collective both backends run well here (gloo's all_gather on device tensors took
    # 1.9 s per call in the 2-rank single-GPU test, its all_reduce 10 ms)
    w, r = world_size(), dist.get_rank()
    out = torch.zeros((w,) + tuple(rows.shape), dtype=rows.dtype, device=rows.device)
    out[r].copy_(rows)
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out.reshape(w * rows.shape[0], *rows.shape[1:])


def allgather(t: torch.Tensor) -> torch.Tensor:
    """Every rank contributes `t` (same shape and dtype everywhere) and receives all of them, [world, *t.shape], rank-major -- the
    exchange of a round of keyframes (descriptors: KBs) and of the dense shards (export / tests).  RCCL: one all_gather.  gloo (CPU
    tests, two ranks on one GPU): the sum-reduce of a zero buffer in which each rank fills its slice -- the bytes are exact either way
    (x + 0 = x; the sign of a zero is not preserved, NaN stays NaN)."""
    w = world_size()
    if w == 1 and not (dist.is_initialized() and FORCE_COLLECTIVES):
        return t[None]
    out = torch.zeros((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, t.contiguous())
        return out
    out[dist.get_rank()].copy_(t)
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out


def share_masks(masks: Optional[torch.Tensor], pixels: int, device, gather=None) -> list:
    """The owner -> replica mask exchange of a round: every rank contributes the masks ITS mask generator produced for
    the keyframe it owns (bool / u8 [n, H, W] on the GPU, n may be 0, `None` = none) and receives every rank's, rank-major = keyframe
    order, as u8 [n_k, pixels] tensors.  Masks travel bit-packed (`ovo_pack_masks` / `ovo_unpack_masks`: 1.2 MB for 32 masks of 640 x 480
    instead of 9.8 MB); two collectives: the counts, then max-count rows of packed words.  Bits are copied, never reduced: what arrives is
    what was sent.  (The seg map is not sent: it is `mask2segmap`'s painting of these same masks, segment_utils.py:12-27.)"""
    from . import _lib as L
    gather = gather or allgather
    lib = L.load()
    if pixels % 16:
        raise L.OvoHipError("share_masks: H * W must be a multiple of 16")
    n = 0 if masks is None else int(masks.shape[0])
    counts = gather(torch.tensor([n], dtype=torch.int32).to(device)).reshape(-1).tolist()
    n_max, words = max(counts), (pixels + 63) // 64
    if n_max == 0:
        return [torch.empty((0, pixels), dtype=torch.uint8, device=device) for _ in counts]
    bits = torch.zeros((n_max, words), dtype=torch.int64, device=device)
    if n:
        m = masks.reshape(n, -1)
        m = L.dev((m.view(torch.uint8) if m.dtype == torch.bool else m).contiguous(), torch.uint8, "masks")
        L.check(lib.ovo_pack_masks(L.ptr(m), n, pixels, L.ptr(bits), words, L.stream()))
    everyone = gather(bits)                                        # [world, n_max, words]
    out = []
    for k, c in enumerate(counts):
        u = torch.empty((c, pixels), dtype=torch.uint8, device=device)
        if c:
            L.check(lib.ovo_unpack_masks(everyone[k].data_ptr(), c, pixels, words, L.ptr(u), L.stream()))
        out.append(u)
    return out


def allreduce_dense_(acc: torch.Tensor, cnt: torch.Tensor, bucket_bytes: int = DENSE_BUCKET_BYTES) -> int:
    """Merge per-GPU dense accumulators (acc f32[N, D], cnt i32[N]) by bucketed in-place sum.  Returns #collectives."""
    if world_size() == 1:
        return 0
    calls = 0
    flat = acc.reshape(-1)
    step = max(1, bucket_bytes // flat.element_size())
    for s in range(0, flat.numel(), step):
        dist.all_reduce(flat[s:s + step], op=dist.ReduceOp.SUM)
        calls += 1
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    return calls + 1


def max_over_ranks(value: float, device) -> float:
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
