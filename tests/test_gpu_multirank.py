"""-m gpu: the frame-sharded multi-GPU path (BASELINE.json configs[3]) must reproduce the one-process run.

Two ranks share ONE GPU (gloo backend, OVO_FORCE_DEVICE=0: RCCL refuses two ranks on a device) and step 4 rounds of 2 keyframes; rank k
owns keyframe k of a round for SAM2 / ViT / pooling, tracking and back-projection run replicated in keyframe order, the round's
descriptors are all-gathered and the dense accumulators are sharded by point.  Everything the map holds afterwards -- points, per-point
instance ids, the instance list, every instance's keyframes / top-k heap / fused descriptor, the per-keyframe descriptor tables, the dense
accumulators, counts, classes and confidences -- must EQUAL the single-process run over the same 8 keyframes (ovo.py:240-282 instance-id
order, vanilla_mapper.py:81-85 append order)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
KW = dict(vit_card="tiny-pe", sam_card="hiera_test", n_map=60_000, n_text=7, scale=0.35, extra_capacity=300_000, track_th=40, k_top_views=3)
N_FRAMES = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _state(pipe):
    n = pipe.slam._n
    acc, cnt, cls, conf = pipe.gather_dense(n)
    ovo = pipe.ovo
    out = {"pcd": pipe.slam.pcd.cpu(), "ids": pipe.slam.pcd_ids.cpu(), "obj_ids": pipe.slam.pcd_obj_ids.cpu(), "colors": pipe.slam.pcd_colors.cpu(),
           "objects": list(ovo.objects), "next_ins_id": ovo.next_ins_id, "table": ovo.get_objs_clips().cpu(),
           "kfs": {i: list(o.kfs_ids) for i, o in ovo.objects.items()}, "top": {i: sorted(o.top_kf) for i, o in ovo.objects.items()},
           "desc": {kf: {i: d.cpu() for i, d in view.items()} for kf, view in ovo.keyframes["ins_descriptors"].items()},
           "acc": acc.cpu(), "cnt": cnt.cpu(), "cls": cls.cpu(), "conf": conf.cpu(),
           "sim": pipe.last["sim"].cpu(), "inst_cls": pipe.last["cls"].cpu()}
    return out


def _source_masks(index):
    """Stand-in for a rank's own mask generator (SAM2 end to end): the masks of keyframe `index`, in mask2segmap order."""
    from ovo_amd import synthetic as syn
    h, w = syn.scannet_depth_hw(0.35)
    e = int(round(syn.SCANNET["crop_edge"] * 0.35))
    m = syn.make_masks(h + 2 * e, w + 2 * e, grid=(3, 4), n_blobs=4, seed=index)
    return torch.from_numpy(syn.masks_to_segmap(m)).to("cuda:0"), torch.from_numpy(m).to("cuda:0")


def _worker(rank, world, port, path, encoder_batch, own_masks=False, n_frames=N_FRAMES):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OVO_FORCE_DEVICE="0", OVO_DIST_BACKEND="gloo")
    from ovo_amd import parallel
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    parallel.init_distributed()
    torch.cuda.set_device(0)
    pipe = FramePipeline("cuda:0", encoder_batch=encoder_batch, **KW)
    assert pipe.world == world and pipe.rank == rank
    frames = synthetic_frames(n_frames, "cuda:0", scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    if own_masks:                                                  # `--sam-full` style: a keyframe's masks exist on its owner only ...
        from ovo_amd.pipeline import Frame
        calls = []
        frames = [Frame(f.index, f.rgb, f.rgb_lr, f.depth, f.c2w, torch.empty(0, device="cuda:0"), torch.empty(0, device="cuda:0")) for f in frames]

        def source(f):
            assert f.index % world == rank                         # ... produced by ITS generator, for its own keyframes only
            calls.append(f.index)
            return _source_masks(f.index)
        pipe.mask_source = source
    for r in range(n_frames // world):
        pipe.step_round(frames[r * world:(r + 1) * world], frames[(r + 1) * world:])
    torch.cuda.synchronize()
    if own_masks:
        assert calls == list(range(rank, n_frames, world)) and pipe.mask_exchanges == n_frames // world
    # invariant of the resident dense map on every shard: it equals a full re-query of the shard's rows
    from ovo_amd.utils import clip_utils
    nl = pipe.local_rows(pipe.slam._n)
    _, cls, conf = clip_utils.similarity(pipe.acc[:nl], pipe.texts, cnt=pipe.cnt[:nl], want_sim=False, want_argmax=True)
    bad = (cls != pipe.dense_cls[:nl]).nonzero().reshape(-1)
    assert bad.numel() == 0, (rank, nl, bad[:8].tolist(), pipe.cnt[:nl][bad[:8]].tolist(), int(pipe.n_touched[0]), int(pipe.n_touched[1]))
    assert torch.equal(conf, pipe.dense_conf[:nl])
    state = _state(pipe)                                           # gather_dense is a collective: every rank calls it
    state["exchanges"] = pipe.exchanges
    state["rows_local"] = pipe.rows_local
    if rank == 0:
        torch.save(state, path)
    parallel.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("encoder_batch,own_masks", [(1, False), (2, False), (2, True)])
def test_two_ranks_reproduce_the_single_process_run(encoder_batch, own_masks):
    """own_masks: every keyframe's masks come from its owner's generator and reach the other rank through `parallel.share_masks`
    (bit-packed all-gather); the run must still equal the one-process run that has all masks locally."""
    _ranks_equal_single_process(2, N_FRAMES, encoder_batch, own_masks)


@pytest.mark.parametrize("encoder_batch,own_masks", [(2, False), (1, True)])
def test_eight_ranks_reproduce_the_single_process_run(encoder_batch, own_masks):
    """BASELINE.json configs[3]'s process layout -- EIGHT ranks, one keyframe each per round -- executed for real: 8 processes share one GPU
    over gloo (the only way a 1-GPU box can run them), 3 rounds = 24 keyframes.  Result rings, descriptor staging (MAX_DESC rows per rank),
    `reserve_round` for 8 map steps, the 8-way block-cyclic point shards and the owner -> replica mask exchange at world 8 must leave exactly
    the state of the one-process run (nothing had executed more than two processes)."""
    _ranks_equal_single_process(8, 24, encoder_batch, own_masks)


def _ranks_equal_single_process(world, n_frames, encoder_batch, own_masks):
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "rank0.pt")
        mp.spawn(_worker, args=(world, _free_port(), path, encoder_batch, own_masks, n_frames), nprocs=world, join=True)
        got = torch.load(path, weights_only=False)
    pipe = FramePipeline("cuda:0", **KW)
    frames = synthetic_frames(n_frames, "cuda:0", scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    for i, f in enumerate(frames):
        pipe.step(f, frames[i + 1:])
    torch.cuda.synchronize()
    ref = _state(pipe)
    assert got["exchanges"] == n_frames // world and got["rows_local"] < ref["acc"].shape[0] + 300_000
    assert len(ref["objects"]) > 5 and ref["cnt"].sum() > 0 and (ref["cls"] >= 0).any(), "fixture too small to mean anything"
    from ovo_amd.utils import clip_utils
    n = pipe.slam._n
    _, full_cls, full_conf = clip_utils.similarity(pipe.acc[:n], pipe.texts, cnt=pipe.cnt[:n], want_sim=False, want_argmax=True)
    assert torch.equal(ref["cls"], full_cls.cpu()) and torch.equal(ref["conf"], full_conf.cpu())      # the reference run's own invariant
    for k in ("pcd", "ids", "obj_ids", "colors", "table", "acc", "cnt", "cls", "conf", "sim", "inst_cls"):
        if not torch.equal(got[k], ref[k]):
            bad = (got[k] != ref[k]).reshape(got[k].shape[0], -1).any(1).nonzero().reshape(-1)
            diff = (got[k].double() - ref[k].double()).abs().reshape(got[k].shape[0], -1).max(1).values[bad]
            raise AssertionError((k, bad.numel(), bad[:16].tolist(), [float(f"{v:.3g}") for v in diff[:16].tolist()],
                                  got[k][bad[:2]].reshape(2, -1)[:, :5].tolist(), ref[k][bad[:2]].reshape(2, -1)[:, :5].tolist()))
    assert got["objects"] == ref["objects"] and got["next_ins_id"] == ref["next_ins_id"]
    assert got["kfs"] == ref["kfs"] and got["top"] == ref["top"]
    assert got["desc"].keys() == ref["desc"].keys()
    for kf in ref["desc"]:
        assert got["desc"][kf].keys() == ref["desc"][kf].keys()
        for i in ref["desc"][kf]:
            assert torch.equal(torch.nan_to_num(got["desc"][kf][i]), torch.nan_to_num(ref["desc"][kf][i])), (kf, i)


def _nccl_worker(rank, world, port, path):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from ovo_amd import parallel
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    parallel.FORCE_COLLECTIVES = True
    out = {}
    t = torch.randn(128, 64, device="cuda:0")
    g = parallel.allgather(t)                                      # dist.all_gather_into_tensor under RCCL
    out["gather_ok"] = bool(g.shape == (1, 128, 64) and torch.equal(g[0], t))
    i = torch.randint(-2 ** 62, 2 ** 62, (7, 33), dtype=torch.int64, device="cuda:0")
    out["gather_i64_ok"] = bool(torch.equal(parallel.allgather(i)[0], i))
    _, masks = _source_masks(3)
    got = parallel.share_masks(masks, masks[0].numel(), torch.device("cuda:0"))
    out["masks_ok"] = bool(len(got) == 1 and torch.equal(got[0].view(torch.bool).reshape(masks.shape), masks))
    out["empty_ok"] = bool(parallel.share_masks(None, masks[0].numel(), torch.device("cuda:0"))[0].shape[0] == 0)
    acc, cnt = torch.ones(1000, 8, device="cuda:0"), torch.ones(1000, dtype=torch.int32, device="cuda:0")
    parallel.FORCE_COLLECTIVES = False
    out["backend"] = dist.get_backend()
    dist.barrier()
    torch.cuda.synchronize()
    torch.save(out, path)
    dist.destroy_process_group()


def test_rccl_branch_of_the_exchange_on_one_gpu():
    """The `nccl` (= RCCL) branch of `parallel.allgather` and the bit-packed mask exchange, executed for real in a one-rank RCCL group
    (RCCL refuses two ranks per device, so the two-rank runs above go through gloo): what comes back is what was sent."""
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "nccl.pt")
        mp.spawn(_nccl_worker, args=(1, _free_port(), path), nprocs=1, join=True)
        out = torch.load(path, weights_only=False)
    assert out["backend"] == "nccl"
    assert out["gather_ok"] and out["gather_i64_ok"] and out["masks_ok"] and out["empty_ok"], out
