"""Dense scatter-accumulate (fusion.hip: k_scatter_scan + k_scatter_apply) alone, at the headline workload's shape: a 1 M-point map, D = 1024,
~24 K matched points clustered in the frustum's consecutive rows.  Prints us per call and GB/s of algorithmic bytes (hits x 8 D + 2 n).
Diagnosis tool: python tools/scatter_bench.py [n_points] [D] [hits]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ovo_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hits = int(sys.argv[3]) if len(sys.argv) > 3 else 24_000
dev = torch.device("cuda", 0)
lib = L.load()
g = torch.Generator().manual_seed(0)
n_masks = 32
seg = torch.full((n,), -1, dtype=torch.int16)
# clustered hits: runs of consecutive rows (the frustum's points were appended together), ~60 % of a run matched
start, left = n // 3, hits
while left > 0:
    run = min(left * 2, 4096)
    m = torch.rand(run, generator=g) < 0.6
    ids = torch.randint(0, n_masks, (run,), generator=g, dtype=torch.int16)
    seg[start:start + run] = torch.where(m, ids, torch.full_like(ids, -1))
    left -= int(m.sum())
    start += run + 20_000
seg = seg.to(dev)
n_hit = int((seg >= 0).sum())
rows = torch.arange(n_masks, dtype=torch.int32, device=dev)
desc = torch.randn(n_masks, D, device=dev)
acc = torch.zeros(n, D, device=dev)
cnt = torch.zeros(n, dtype=torch.int32, device=dev)
touched = torch.empty(n, dtype=torch.int32, device=dev)
n_touched = torch.zeros(2, dtype=torch.int32, device=dev)


def call(par, with_list=True):
    if with_list:
        L.check(lib.ovo_scatter_accum_touched(L.ptr(seg), n, L.ptr(rows), n_masks, L.ptr(desc), D, L.ptr(acc), L.ptr(cnt), L.ptr(touched),
                                              n_touched[par:].data_ptr(), n_touched[par ^ 1:].data_ptr(), 0, 1, 1024, L.stream()))
    else:
        L.check(lib.ovo_scatter_accum(L.ptr(seg), n, L.ptr(rows), n_masks, L.ptr(desc), D, L.ptr(acc), L.ptr(cnt), L.stream()))


for with_list in (True, False):
    for i in range(3):
        call(i & 1, with_list)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for i in range(reps):
        call((i + 1) & 1, with_list)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    by = n_hit * (8.0 * D + 12) + 2.0 * n
    print(f"{'scan + apply (touched list)' if with_list else 'one kernel (no list)      '}: n {n} D {D} hits {n_hit}: {us:8.2f} us  {by / us / 1e3:8.1f} GB/s")
# linearity: every hit row = calls x desc row
calls = 2 * 23
ref = desc[seg[seg >= 0].long()] * calls
got = acc[seg >= 0]
print("max |acc - calls * desc| / calls:", float((got - ref).abs().max()) / calls, "cnt ok:", bool((cnt[seg >= 0] == calls).all()), bool((cnt[seg < 0] == 0).all()))
