"""The measured tile table (ovo_amd/csrc/gemm_tuned.h, written by tools/gemm_tune.py): well-formed on the host; on the GPU every listed product gives the
cost model's result (the table only changes which kernel family runs a shape -- all families accumulate an output element in the same k-order)."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT

TILES = {"256x256", "256x128", "128x128", "128x64", "64x128", "64x64", "stream"}


def _entries():
    src = open(os.path.join(ROOT, "ovo_amd", "csrc", "gemm_tuned.h")).read()
    body = src[src.index("kTunedTiles[] = {"):]
    return [(int(m), int(n), int(k), int(f), t) for m, n, k, f, t in re.findall(r'\{(\d+), (\d+), (\d+), (\d+), "([0-9a-z]+)"\}', body)], body


def test_table_is_well_formed():
    ent, body = _entries()
    assert "{0, 0, 0, 0, nullptr}" in body                          # the terminator the dispatcher's loop relies on (e.tile == nullptr never matches)
    assert len({e[:4] for e in ent}) == len(ent)                    # one choice per (M, N, K, variant)
    for m, n, k, f, t in ent:
        assert t in TILES and m > 0 and n > 0 and k > 0 and k % 32 == 0
        assert f & ~(1 | 2 | 12 | 16) == 0                          # only variants tools/gemm_tune.py can reproduce: f32 out, residual, activation, rotary
        if t == "stream":
            assert m >= 16384 and k <= 256


@pytest.mark.gpu
def test_tuned_choice_equals_cost_model_choice(monkeypatch):
    from ovo_amd import _lib as L
    ent, _ = _entries()
    if not ent:
        pytest.skip("empty table")
    lib = L.load()
    monkeypatch.setenv("OVO_GELU_POLY", "1")                        # one GELU form in every family (the ring kernels have no LDS table)
    g0 = torch.Generator().manual_seed(11)
    for m, n, k, f, t in sorted(ent, key=lambda e: e[0] * e[1] * e[2])[:16]:     # the sixteen smallest: seconds, every family among them
        a = torch.randn(m, k, generator=g0).to(torch.bfloat16).cuda()
        w = (torch.randn(n, k, generator=g0) * k ** -0.5).to(torch.bfloat16).cuda()
        bias = torch.randn(n, generator=g0).cuda()
        f32 = bool(f & 1)
        res = torch.randn(m, n, generator=g0).cuda() if f & 2 else None
        outs = []
        T, hd = 577, 64
        cs, sn = torch.rand(T, hd, generator=g0).cuda(), torch.rand(T, hd, generator=g0).cuda()       # (rotary entries: ONE table for both runs)
        for tuned in (True, False):
            if tuned:
                monkeypatch.delenv("OVO_GEMM_NO_TUNED", raising=False)
            else:
                monkeypatch.setenv("OVO_GEMM_NO_TUNED", "1")
            out = res.clone() if (res is not None and f32) else torch.zeros(m, n, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
            g = L.Gemm()
            g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc = a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), out.data_ptr(), n
            g.add, g.ld_add = (out.data_ptr() if f32 else res.data_ptr(), n) if res is not None else (None, 0)
            g.M, g.N, g.K, g.in_dtype, g.out_dtype, g.act, g.alpha = m, n, k, 2, 0 if f32 else 2, (f & 12) >> 2, 1.0
            if f & 16:
                rope = L.Rope(); rope.cos, rope.sin, rope.T, rope.hd, rope.cols, rope.t0 = cs.data_ptr(), sn.data_ptr(), T, hd, 2 * n // 3, 1
                L.check(lib.ovo_gemm_rope(C.byref(g), C.byref(rope), L.stream()))
            else:
                L.check(lib.ovo_gemm(C.byref(g), L.stream()))
            torch.cuda.synchronize()
            outs.append(out)
        diff = (outs[0].float() - outs[1].float()).abs()
        assert torch.equal(outs[0], outs[1]), (m, n, k, f, t, "max |difference|", diff.max().item(), "elements", int((diff > 0).sum()))
        if not f & 16:
            ref = a.float() @ w.float().T + bias
            if f & 12:
                ref = torch.nn.functional.gelu(ref)
            if res is not None:
                ref = ref + res
            torch.testing.assert_close(outs[0].float(), ref, atol=0.06, rtol=0.02)
