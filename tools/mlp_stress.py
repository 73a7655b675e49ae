"""Determinism stress of the fused MLP kernel (mlp_stream.hip): repeated launches on one input, with random allocations (shifting addresses) and a
GEMM between them, compared bit for bit with the first launch.  Variant 1 (one 512-thread workgroup per CU) is the production form; variants 2 / 3
(two 256-thread workgroups per CU) are reachable for diagnosis and FAIL this (mlp_stream_launch).  OVO_MLP_DBG: 1 = extra barrier per chunk,
2 = wait for every DMA right after its issue.   python tools/mlp_stress.py"""
import ctypes as C, os, sys, random
os.environ["OVO_KNOBS_DYNAMIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
DEV = "cuda:0"
lib = L.load()
def stress(rows, d, k1, variant, dbg, iters=400, perturb=True):
    os.environ["OVO_MLP_RB"] = str(variant); os.environ["OVO_MLP_DBG"] = str(dbg)
    hid = 4 * d
    g = torch.Generator().manual_seed(rows + d)
    x0 = (torch.randn(rows, d, generator=g) * 2 + 0.5).to(DEV)
    gamma, beta = (torch.randn(d, generator=g) * 0.5 + 1).to(DEV), (torch.randn(d, generator=g) * 0.1).to(DEV)
    w1 = torch.zeros(hid, k1, dtype=torch.bfloat16, device=DEV)
    w1[:, :d] = (torch.randn(hid, d, generator=g) * d ** -0.5).to(DEV, torch.bfloat16)
    w2 = (torch.randn(d, hid, generator=g) * hid ** -0.5).to(DEV, torch.bfloat16)
    b1, b2 = torch.randn(hid, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    def call(xf):
        rc = lib.ovo_mlp_f32(xf.data_ptr(), rows, d, gamma.data_ptr(), beta.data_ptr(), 1e-6, w1.data_ptr(), k1, b1.data_ptr(), hid, w2.data_ptr(), hid, b2.data_ptr(), L.stream())
        assert rc == 0, rc
    ref = x0.clone(); call(ref); torch.cuda.synchronize()
    rnd = random.Random(1)
    big = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    bad, worst, junk = 0, 0.0, []
    for it in range(iters):
        if perturb:
            junk.append(torch.empty(rnd.randrange(1, 1 << 22), dtype=torch.uint8, device=DEV))
            if len(junk) > 6: junk.pop(rnd.randrange(len(junk)))
            if it % 3 == 0: big @ big
        xf = torch.cat([x0, torch.full((rnd.randrange(1, 64), d), 7.0, device=DEV)]) if perturb else x0.clone()
        call(xf)
        if it % 4 == 0 or not perturb: torch.cuda.synchronize()
        if not torch.equal(xf[:rows], ref):
            bad += 1
            dd = (xf[:rows] - ref).abs(); worst = max(worst, float(dd.max()))
            if bad <= 3:
                b = (dd > 0).nonzero()
                print("      it", it, "differing", b.shape[0], "row blocks", sorted(set((b[:, 0] // 16).tolist()))[:8], "rows%16", sorted(set((b[:, 0] % 16).tolist()))[:16], "cols", sorted(set(b[:, 1].tolist()))[:20], "max", float(dd.max()))
    print(f"variant {variant} dbg {dbg} ({rows},{d}) perturb {perturb}: {bad} of {iters} launches differ from the first, worst {worst:.3e}")
for v, dbg in ((2, 0), (1, 0), (2, 1), (2, 2), (3, 0)):
    stress(65536, 112, 128, v, dbg)
stress(32768 + 40, 224, 256, 2, 0); stress(32768 + 40, 224, 256, 1, 0)
stress(786432, 112, 128, 2, 0, iters=60); stress(786432, 112, 128, 1, 0, iters=60)
