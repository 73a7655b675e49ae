"""Dense scatter-reduce fusion alone, at the headline workload's shape: a 1 M-point map (the accumulator is sized for SETS disjoint frustums), D = 1024,
~24 K matched points clustered in consecutive rows, 10 texts.  Three forms:
    fused      ovo_scatter_accum_query: ONE launch from a hit list (what the tracking pass emits): accumulate + re-query of the changed rows (round 6)
    3 launches ovo_scatter_accum_touched (scan + apply) + ovo_similarity_rows: the round-5 path
    scan+apply ovo_scatter_accum_touched alone (the number round 5 quoted)
Each is timed WARM (the same 24 K rows every call: 100 MB of accumulator rows + ids stay in the 256 MB Infinity Cache -- what round 5's "4.6 TB/s" was) and
COLD (SETS = 12 disjoint hit sets visited in turn: 1.2 GB of other rows pass between two visits of a row -- what a keyframe sees in the pipeline, where
consecutive keyframes touch different rows behind 2 ms of encoder traffic).  VERDICT r5 item 3a.   python tools/scatter_bench.py [n_points] [D] [hits]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ovo_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hits = int(sys.argv[3]) if len(sys.argv) > 3 else 24_000
SETS, Q = 12, 10
dev = torch.device("cuda", 0)
lib = L.load()
g = torch.Generator().manual_seed(0)
n_masks = 32
rows = torch.arange(n_masks, dtype=torch.int32, device=dev)
desc = torch.randn(n_masks, D, device=dev)
texts = torch.nn.functional.normalize(torch.randn(Q, D, device=dev), dim=1)
acc = torch.zeros(n, D, device=dev)
cnt = torch.zeros(n, dtype=torch.int32, device=dev)
cls = torch.zeros(n, dtype=torch.int64, device=dev)
conf = torch.zeros(n, dtype=torch.float32, device=dev)
touched = torch.empty(n, dtype=torch.int32, device=dev)
n_touched = torch.zeros(2, dtype=torch.int32, device=dev)
segs, lists, counts = [], [], []
span = n // SETS
for s in range(SETS):                                             # clustered hits: runs of consecutive rows (~60 % of a run matched), one region per set
    seg = torch.full((n,), -1, dtype=torch.int16)
    start, left = s * span + 1000, hits
    while left > 0 and start + 4096 < (s + 1) * span:
        run = min(left * 2, 4096)
        m = torch.rand(run, generator=g) < 0.6
        ids = torch.randint(0, n_masks, (run,), generator=g, dtype=torch.int16)
        seg[start:start + run] = torch.where(m, ids, torch.full_like(ids, -1))
        left -= int(m.sum())
        start += run + 2_000
    idx = torch.nonzero(seg >= 0).flatten().to(torch.int32)
    idx = idx[torch.randperm(idx.numel(), generator=g)]             # the tracking pass's list comes in no particular order
    segs.append(seg.to(dev)); lists.append(torch.cat([idx, torch.tensor([idx.numel(), 0, 0, 0], dtype=torch.int32)]).to(dev)); counts.append(int(idx.numel()))
par = [0]


def fused(s):
    h = lists[s]
    L.check(lib.ovo_scatter_accum_query(L.ptr(h), h[counts[s]:].data_ptr(), n, L.ptr(segs[s]), L.ptr(rows), n_masks, L.ptr(desc), D, L.ptr(acc), L.ptr(cnt), 0, 1, 1024,
                                        L.ptr(texts), Q, 0, 0.0, 0.0, 0.0, L.ptr(cls), L.ptr(conf), L.stream()))


def scan_apply(s):
    p = par[0]; par[0] ^= 1
    L.check(lib.ovo_scatter_accum_touched(L.ptr(segs[s]), n, L.ptr(rows), n_masks, L.ptr(desc), D, L.ptr(acc), L.ptr(cnt), L.ptr(touched), n_touched[p:].data_ptr(),
                                          n_touched[p ^ 1:].data_ptr(), 0, 1, 1024, L.stream()))
    return p


def three(s):
    p = scan_apply(s)
    L.check(lib.ovo_similarity_rows(L.ptr(acc), 0, L.ptr(touched), n_touched[p:].data_ptr(), n, D, L.ptr(texts), Q, L.ptr(cnt), 0, 0.0, 0.0, 0.0, L.ptr(cls), L.ptr(conf),
                                    L.stream()))


def time_it(fn, cold, reps=24):
    for i in range(SETS):
        fn(i if cold else 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % SETS if cold else 0)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


n_hit = sum(counts) / SETS
print(f"n {n} D {D} hits per call {n_hit:.0f} ({SETS} disjoint sets, {n_hit * D * 4 / 1e6:.0f} MB of rows each); algorithmic bytes per call = hits x (8 D + 12): "
      f"{n_hit * (8 * D + 12) / 1e6:.1f} MB (+ 2 n for the forms that scan point_seg)")
for name, fn, scan in (("fused accumulate + re-query (1 launch)", fused, 0), ("scan + apply + similarity_rows (3 launches)", three, 1), ("scan + apply only (2 launches)", scan_apply, 1)):
    for cold in (True, False):
        us = time_it(fn, cold)
        by = n_hit * (8.0 * D + 12) + (2.0 * n if scan else 0.0)
        print(f"{name:46s} {'COLD (rotating sets)' if cold else 'warm (same rows)    '}: {us:8.2f} us  {by / us / 1e3:8.1f} GB/s of algorithmic bytes")
