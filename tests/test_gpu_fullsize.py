"""-m gpu: BASELINE.json's full sizes (5 M / 10 M-point maps, 1k texts) through size-independent properties -- the oracle
finishes in seconds only up to ~1 M points (tests/test_gpu_geometry.py covers that range against it bit for bit)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _big_map(n, seed):
    """n points: the synthetic room tiled with jitter, generated on the device (host generation would dominate the test)."""
    from ovo_amd import synthetic as syn
    base = torch.from_numpy(syn.padded_map(1_000_000, frames=3, scale=1.0, seed=seed)).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(seed)
    reps = (n + base.shape[0] - 1) // base.shape[0]
    pts = base.repeat(reps, 1)[:n].clone()
    pts += (torch.rand(pts.shape, generator=g, device=DEV) - 0.5) * 0.02
    return pts.contiguous()


@pytest.mark.parametrize("n", [5_000_000, 10_000_000])
def test_tracking_pass_properties_full_size(n):
    """Fused cull + project + match + seg lookup + vote over the whole map:
       (1) the same per-point answer as the unfused frustum / match kernels, (2) votes are additive over a split of the map,
       (3) match indices ascending and unique, (4) assignment idempotent."""
    from oracle import geometry as OG
    from ovo_amd import _lib as L, synthetic as syn
    from ovo_amd.utils import geometry_utils as G
    lib = L.load()
    h, w = syn.scannet_depth_hw(1.0)
    K = syn.scannet_intrinsics(1.0)
    c2w = syn.pose(2)
    depth = syn.render_depth(c2w, K, h, w, seed=5)
    H, W = 480, 640
    masks = syn.make_masks(H, W, seed=9)
    seg = syn.masks_to_segmap(masks)
    pts = _big_map(n, seed=5)
    g = torch.Generator(device=DEV).manual_seed(1)
    ins = torch.where(torch.rand(n, generator=g, device=DEV) < 0.3, torch.full((n,), -1, device=DEV, dtype=torch.int32),
                      torch.randint(0, 500, (n,), generator=g, device=DEV, dtype=torch.int32)).contiguous()
    corners = OG.frustum_corners(depth, c2w, K)
    w2c = torch.linalg.inv(torch.from_numpy(c2w))
    cam = G.make_camera(torch.from_numpy(corners), w2c, torch.from_numpy(K), 0.05, h, w)
    n_masks, cols = masks.shape[0], 501
    d_depth, d_seg = torch.from_numpy(depth).to(DEV), torch.from_numpy(seg).to(DEV)

    def run(p, i):
        m = p.shape[0]
        ps = torch.empty(m, dtype=torch.int16, device=DEV)
        hist = torch.empty((n_masks, cols), dtype=torch.int32, device=DEV)
        cnt = torch.empty(2, dtype=torch.int64, device=DEV)
        L.check(lib.ovo_track_project(L.ptr(p), L.ptr(i), m, cam, L.ptr(d_depth), L.ptr(d_seg), H, W, L.Ratio(1, 1.0, 1.0, 12), L.ptr(ps),
                                      L.ptr(hist), n_masks, cols, L.ptr(cnt), L.stream()))
        return ps, hist, cnt
    ps, hist, cnt = run(pts, ins)
    half = n // 2 + 17
    ps_a, hist_a, cnt_a = run(pts[:half], ins[:half])
    ps_b, hist_b, cnt_b = run(pts[half:].contiguous(), ins[half:].contiguous())
    assert torch.equal(ps, torch.cat([ps_a, ps_b])) and torch.equal(hist, hist_a + hist_b) and torch.equal(cnt, cnt_a + cnt_b)
    # unfused kernels: frustum ids, then match on the gathered points
    ids = G.compute_frustum_point_ids(pts, torch.from_numpy(corners), device=DEV)
    assert ids.shape[0] == int(cnt[0]) and bool((ids[1:] > ids[:-1]).all())
    mi, muv = G.match_3d_points_to_2d_pixels(d_depth, w2c, pts.index_select(0, ids), torch.from_numpy(K), 0.05)
    assert mi.shape[0] == int(cnt[1]) and bool((mi[1:] > mi[:-1]).all()) and mi.shape[0] > 10_000
    matched = ids.index_select(0, mi)
    assert torch.equal(torch.nonzero(ps >= -1).reshape(-1), matched)
    uv = muv.long() + 12
    assert torch.equal(ps[matched].int(), d_seg[uv[:, 1], uv[:, 0]])
    assert int(hist.sum()) == int((ps >= 0).sum())
    # assignment: idempotent, touches only free points of targeted masks
    target = torch.where(torch.arange(n_masks) % 2 == 0, 9000 + torch.arange(n_masks), torch.tensor(-1)).to(DEV, torch.int32)
    out1, out2 = torch.empty_like(ins), torch.empty_like(ins)
    L.check(lib.ovo_assign_instances(L.ptr(ins), L.ptr(ps), n, L.ptr(target), n_masks, L.ptr(out1), None, L.stream()))
    L.check(lib.ovo_assign_instances(L.ptr(out1), L.ptr(ps), n, L.ptr(target), n_masks, L.ptr(out2), None, L.stream()))
    assert torch.equal(out1, out2)
    changed = out1 != ins
    assert bool((ins[changed] == -1).all()) and bool((ps[changed] >= 0).all()) and int(changed.sum()) > 0


def test_dense_fusion_5m_points_linearity():
    """Config 4's map size: per-point accumulate of mask descriptors.  acc is linear in the number of passes, counts are exact,
    untouched rows stay zero, and the mean descriptor queried afterwards is the descriptor itself."""
    from ovo_amd import _lib as L
    from ovo_amd.utils import clip_utils as CU
    n, D, n_masks = 5_000_000, 1024, 48
    g = torch.Generator(device=DEV).manual_seed(7)
    seg = torch.randint(-2, n_masks, (n,), generator=g, device=DEV, dtype=torch.int16)
    mask_row = torch.where(torch.arange(n_masks) % 5 == 0, torch.tensor(-1), torch.randperm(n_masks)).to(DEV, torch.int32)
    desc = torch.nn.functional.normalize(torch.randn((n_masks, D), generator=g, device=DEV), dim=1)
    acc = torch.zeros((n, D), dtype=torch.float32, device=DEV)
    cnt = torch.zeros(n, dtype=torch.int32, device=DEV)
    lib = L.load()
    for _ in range(2):
        L.check(lib.ovo_scatter_accum(L.ptr(seg), n, L.ptr(mask_row), n_masks, L.ptr(desc), D, L.ptr(acc), L.ptr(cnt), L.stream()))
    rows = torch.where(seg >= 0, mask_row[seg.clamp(min=0).long()], torch.tensor(-1, device=DEV, dtype=torch.int32))
    hit = rows >= 0
    assert torch.equal(cnt, 2 * hit.int())
    sample = torch.randint(0, n, (200_000,), generator=g, device=DEV)
    want = torch.where(hit[sample][:, None], 2 * desc[rows[sample].clamp(min=0).long()], torch.zeros((1, D), device=DEV))
    assert torch.equal(acc[sample], want)                                   # x + x is exact
    assert float(acc.abs().sum(1)[~hit].max()) == 0.0
    # query of the mean descriptors with the descriptors themselves as "texts": every touched point scores 1 on its own mask
    sim_cls = CU.similarity(acc, desc, cnt=cnt, want_sim=False, want_argmax=True, th=0.5)
    assert torch.equal(sim_cls[1][hit], rows[hit].long()) and bool((sim_cls[1][~hit] == -1).all())
    assert float((sim_cls[2][hit] - 1.0).abs().max()) < 1e-5


def test_dense_query_10m_points_1k_texts():
    """Config 5's shape: fused-map descriptors (fp16) x 1000 texts.  Rows are independent, so the whole map must give exactly
    what its chunks give; a random sample is checked against an fp32 matmul (1e-3: the north-star tolerance)."""
    from ovo_amd.utils import clip_utils as CU
    n, d, q = 10_000_000, 768, 1000
    g = torch.Generator(device=DEV).manual_seed(3)
    F = torch.empty((n, d), dtype=torch.float16, device=DEV)
    for s in range(0, n, 1_000_000):                       # generate in slabs: no 30 GB fp32 temporary
        x = torch.randn((min(1_000_000, n - s), d), generator=g, device=DEV)
        F[s:s + x.shape[0]] = torch.nn.functional.normalize(x, dim=1).half()
    T = torch.nn.functional.normalize(torch.randn((q, d), generator=g, device=DEV), dim=1)
    _, cls, conf = CU.similarity(F, T, want_sim=False, want_argmax=True)
    assert cls.shape == (n,) and int(cls.min()) >= 0 and int(cls.max()) < q
    for s, e in ((0, 1_250_000), (6_000_003, 7_250_003), (n - 999_999, n)):
        _, c2, f2 = CU.similarity(F[s:e], T, want_sim=False, want_argmax=True)
        assert torch.equal(c2, cls[s:e]) and torch.equal(f2, conf[s:e])
    idx = torch.randint(0, n, (4096,), generator=g, device=DEV)
    ref = F[idx].float() @ T.half().float().t()
    rmax, rarg = ref.max(1)
    assert float((conf[idx] - rmax).abs().max()) < 1e-3
    agree = cls[idx] == rarg
    assert float(agree.float().mean()) > 0.999             # a near-tie may resolve differently in fp32 accumulation order
    assert float((ref.gather(1, cls[idx][:, None]).squeeze(1) - rmax).abs().max()) < 1e-3


@pytest.mark.parametrize("m,n,k,act", [(524288, 448, 128, 1), (524288, 336, 128, 0), (131072, 896, 256, 1), (524288, 576, 192, 0)])
def test_streaming_gemm_full_size_equals_tiled(monkeypatch, m, n, k, act):
    """The Hiera stage-1 / stage-2 products at the bench's size (8 frames x 65 536 / 16 384 tokens): the weights-resident streaming kernel
    and the tiled kernels accumulate every output element in the same k-order, so the full-size outputs are bit-identical (with GELU: equal up to an ulp of bf16 on < 2 % of the elements -- table vs polynomial); a sampled set
    of rows is also checked against the fp32 product of the same rounded operands."""
    import ctypes as C
    from ovo_amd import _lib as L
    g = torch.Generator().manual_seed(m % 1000 + n + k)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    outs = []
    for mode in ("tiled", "stream"):
        if mode == "tiled":
            monkeypatch.setenv("OVO_GEMM_NO_STREAM", "1")
        else:
            monkeypatch.delenv("OVO_GEMM_NO_STREAM")
            monkeypatch.setenv("OVO_GEMM_TILE", "stream")
        out = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
        gg = L.Gemm()
        gg.A, gg.lda, gg.W, gg.ldw, gg.bias, gg.C, gg.ldc, gg.add, gg.ld_add = a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), out.data_ptr(), n, None, 0
        gg.M, gg.N, gg.K, gg.in_dtype, gg.out_dtype, gg.act, gg.alpha = m, n, k, 2, 2, act, 1.0
        L.check(L.load().ovo_gemm(C.byref(gg), L.stream()))
        outs.append(out)
    if act:       # GELU: the streaming kernel's LDS table against the tiled kernel's own form on the same pre-activation bits -- the bf16 outputs differ
        # by an ulp where a rounding boundary falls between two 1e-5-accurate approximations, nowhere by more
        d = (outs[0].float() - outs[1].float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(outs[0].float().abs().max()) and float((d > 0).float().mean()) < 0.02
    else:
        assert torch.equal(outs[0], outs[1])
    rows = torch.randint(0, m, (2048,), generator=g).to(DEV)
    ref = a[rows].float() @ w.float().T + bias
    if act:
        ref = torch.nn.functional.gelu(ref)
    torch.testing.assert_close(outs[1][rows].float(), ref, atol=0.03, rtol=0.01)
