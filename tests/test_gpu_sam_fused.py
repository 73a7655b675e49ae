"""-m gpu: the fused image-side kernels of the SAM2 mask decoder (csrc/samfuse.hip) one by one, each against a plain torch fp32
restatement of the same arithmetic on the same bf16-rounded operands (sam2 MaskDecoder / TwoWayAttentionBlock reached at
segment_utils.py:291-308) and against the unfused chain of this library (ovo_gemm + row pass) it replaces."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lib():
    from ovo_amd import _lib as L
    return L, L.load()


def _bf(t):
    return t.to(torch.bfloat16)


def _gelu(x):
    return torch.nn.functional.gelu(x)


@pytest.mark.parametrize("N,K", [(256, 128), (128, 64)])
@pytest.mark.parametrize("res_kind", ["periodic_f32", "full_f32", "full_bf16", "none"])
def test_proj_ln_vs_torch(N, K, res_kind):
    L, lib = _lib()
    g = torch.Generator().manual_seed(N + K)
    S, P = 48, 5                                                   # M = 240 rows: not a multiple of 16 x 8 waves
    M = S * P
    A = _bf(torch.randn(M, K, generator=g)).to(DEV)
    W = _bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    pe = torch.randn(S, N, generator=g).to(DEV)
    res = res16 = None
    rows = 0
    if res_kind == "periodic_f32":
        res, rows = torch.randn(S, N, generator=g).to(DEV), S
    elif res_kind == "full_f32":
        res, rows = torch.randn(M, N, generator=g).to(DEV), M
    elif res_kind == "full_bf16":
        res16, rows = _bf(torch.randn(M, N, generator=g)).to(DEV), M
    y32 = torch.empty(M, N, device=DEV)
    y16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ype = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L.check(lib.ovo_sam_proj_ln(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(res), L.ptr(res16), max(rows, 1), L.ptr(gamma), L.ptr(beta), 1e-5, L.ptr(pe), S,
                                L.ptr(y32), L.ptr(y16), L.ptr(ype), M, N, K, L.stream()))
    x = A.float() @ W.float().T + bias
    if res is not None:
        x = x + res.repeat(M // rows, 1)
    if res16 is not None:
        x = x + res16.float()
    ref = torch.nn.functional.layer_norm(x, (N,), gamma, beta, 1e-5)
    torch.testing.assert_close(y32, ref, atol=2e-4, rtol=1e-4)
    assert torch.equal(y16, _bf(y32))                              # the bf16 copy is the rounding of the f32 result
    assert torch.equal(ype, _bf(y32 + pe.repeat(P, 1)))


def test_proj_ln_reports_unsupported_widths():
    L, lib = _lib()
    t = torch.zeros(64, 96, dtype=torch.bfloat16, device=DEV)
    f = torch.zeros(96, device=DEV)
    o = torch.empty(64, 96, dtype=torch.bfloat16, device=DEV)
    rc = lib.ovo_sam_proj_ln(L.ptr(t), L.ptr(t), None, None, None, 1, L.ptr(f), L.ptr(f), 1e-5, None, 1, None, L.ptr(o), None, 64, 96, 96, L.stream())
    assert rc == L.E_UNSUPPORTED


@pytest.mark.parametrize("N,K", [(256, 256), (128, 256), (128, 128), (64, 128)])
def test_skinny_linear_vs_torch_and_periodic_gemm(N, K):
    L, lib = _lib()
    g = torch.Generator().manual_seed(3 * N + K)
    S, P, NTOT = 40, 7, N + 64                                     # the product is a column block of a wider output / add matrix
    M = S * P
    A = _bf(torch.randn(M, K, generator=g)).to(DEV)
    W = _bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    add = torch.randn(S, NTOT, generator=g).to(DEV)
    out = torch.full((M, NTOT), 7.0, dtype=torch.bfloat16, device=DEV)
    off = 32
    L.check(lib.ovo_sam_linear(L.ptr(A), L.ptr(W), L.ptr(bias), C.c_void_p(add.data_ptr() + 4 * off), S, NTOT, C.c_void_p(out.data_ptr() + 2 * off),
                               NTOT, M, N, K, L.stream()))
    ref = A.float() @ W.float().T + bias + add[:, off:off + N].repeat(P, 1)
    torch.testing.assert_close(out[:, off:off + N].float(), ref, atol=0.03, rtol=0.01)           # bf16 output
    assert torch.all(out[:, :off] == 7.0) and torch.all(out[:, off + N:] == 7.0)                 # neighbours untouched
    # the tiled-GEMM form of the same product (ovo_gemm_periodic): equal up to the summation order of the f32 accumulation
    gm = L.Gemm()
    out2 = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    addc = add[:, off:off + N].contiguous()
    gm.A, gm.lda, gm.W, gm.ldw, gm.bias, gm.C, gm.ldc, gm.add, gm.ld_add = A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out2.data_ptr(), N, addc.data_ptr(), N
    gm.M, gm.N, gm.K, gm.in_dtype, gm.out_dtype, gm.act, gm.alpha = M, N, K, 2, 2, 0, 1.0
    L.check(lib.ovo_gemm_periodic(C.byref(gm), S, L.stream()))
    torch.testing.assert_close(out2.float(), out[:, off:off + N].float(), atol=0.02, rtol=0.01)
    torch.testing.assert_close(out2.float(), ref, atol=0.03, rtol=0.01)


@pytest.mark.parametrize("C1,K,s,P", [(64, 256, 8, 3), (32, 128, 6, 5)])
def test_up1_ln_vs_torch_and_unfused(C1, K, s, P):
    L, lib = _lib()
    g = torch.Generator().manual_seed(C1 + s)
    M = P * s * s
    A = _bf(torch.randn(M, K, generator=g)).to(DEV)
    W = _bf(torch.randn(4 * C1, K, generator=g) * K ** -0.5).to(DEV)                              # row = (dy * 2 + dx) * C1 + c
    bias = (0.1 * torch.randn(C1, generator=g)).to(DEV)
    feat = (0.5 * torch.randn(2 * s, 2 * s, C1, generator=g)).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(C1, generator=g)).to(DEV), (0.1 * torch.randn(C1, generator=g)).to(DEV)
    out = torch.empty(P, 2 * s, 2 * s, C1, dtype=torch.bfloat16, device=DEV)
    L.check(lib.ovo_sam_up1_ln(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(feat), L.ptr(gamma), L.ptr(beta), 1e-6, P, s, C1, K, L.ptr(out), L.stream()))
    prod = (A.float() @ W.float().T).reshape(P, s, s, 2, 2, C1).permute(0, 1, 3, 2, 4, 5).reshape(P, 2 * s, 2 * s, C1)    # pixel shuffle
    ref = _gelu(torch.nn.functional.layer_norm(prod + bias + feat, (C1,), gamma, beta, 1e-6))
    torch.testing.assert_close(out.float(), ref, atol=0.03, rtol=0.01)
    # the unfused chain rounds the product to bf16 before the row pass; the results agree to bf16 resolution
    gm = L.Gemm()
    g1 = torch.empty((M, 4 * C1), dtype=torch.bfloat16, device=DEV)
    gm.A, gm.lda, gm.W, gm.ldw, gm.bias, gm.C, gm.ldc, gm.add, gm.ld_add = A.data_ptr(), K, W.data_ptr(), K, None, g1.data_ptr(), 4 * C1, None, 0
    gm.M, gm.N, gm.K, gm.in_dtype, gm.out_dtype, gm.act, gm.alpha = M, 4 * C1, K, 2, 2, 0, 1.0
    L.check(lib.ovo_gemm(C.byref(gm), L.stream()))
    out2 = torch.empty_like(out)
    L.check(lib.ovo_sam_upscale_ln(L.ptr(g1), L.ptr(bias), L.ptr(feat), L.ptr(gamma), L.ptr(beta), 1e-6, P, s, C1, L.ptr(out2), L.stream()))
    torch.testing.assert_close(out2.float(), out.float(), atol=0.05, rtol=0.02)


@pytest.mark.parametrize("C2,K,s2,P,n_mask,first", [(32, 64, 8, 3, 4, 1), (16, 32, 6, 5, 4, 0), (32, 64, 4, 2, 1, 0)])
def test_up2_masks_vs_torch_and_unfused(C2, K, s2, P, n_mask, first):
    L, lib = _lib()
    g = torch.Generator().manual_seed(C2 + s2 + n_mask)
    M = P * s2 * s2
    A = _bf(torch.randn(M, K, generator=g)).to(DEV)
    W = _bf(torch.randn(4 * C2, K, generator=g) * K ** -0.5).to(DEV)
    bias = (0.1 * torch.randn(C2, generator=g)).to(DEV)
    feat = (0.5 * torch.randn(2 * s2, 2 * s2, C2, generator=g)).to(DEV)
    hyper = torch.randn(P, n_mask, C2, generator=g).to(DEV)
    n_out = n_mask - first
    out = torch.empty(P, n_out, 2 * s2, 2 * s2, device=DEV)
    L.check(lib.ovo_sam_up2_masks(L.ptr(A), L.ptr(W), L.ptr(bias), L.ptr(feat), L.ptr(hyper), n_mask, first, P, s2, C2, K, L.ptr(out), L.stream()))
    prod = (A.float() @ W.float().T).reshape(P, s2, s2, 2, 2, C2).permute(0, 1, 3, 2, 4, 5).reshape(P, 2 * s2, 2 * s2, C2)
    up = _gelu(prod + bias + feat)
    ref = torch.einsum("pyxc,pmc->pmyx", up, hyper[:, first:])
    torch.testing.assert_close(out, ref, atol=2e-3 * C2 ** 0.5, rtol=1e-3)
    gm = L.Gemm()
    g2 = torch.empty((M, 4 * C2), dtype=torch.bfloat16, device=DEV)
    gm.A, gm.lda, gm.W, gm.ldw, gm.bias, gm.C, gm.ldc, gm.add, gm.ld_add = A.data_ptr(), K, W.data_ptr(), K, None, g2.data_ptr(), 4 * C2, None, 0
    gm.M, gm.N, gm.K, gm.in_dtype, gm.out_dtype, gm.act, gm.alpha = M, 4 * C2, K, 2, 2, 0, 1.0
    L.check(lib.ovo_gemm(C.byref(gm), L.stream()))
    out2 = torch.empty_like(out)
    L.check(lib.ovo_sam_upscale_masks(L.ptr(g2), L.ptr(bias), L.ptr(feat), L.ptr(hyper), n_mask, first, P, s2, C2, L.ptr(out2), L.stream()))
    torch.testing.assert_close(out2, out, atol=0.05 * C2 ** 0.5, rtol=0.02)                       # the unfused chain rounds the product to bf16


@pytest.mark.parametrize("H,T,S,P,shared,fusedcols", [(8, 8, 4096, 3, False, 3), (8, 8, 1000, 2, True, 1), (4, 8, 333, 5, False, 2), (2, 8, 64, 1, False, 1)])
def test_t2i_attention_vs_torch_and_flash_kernel(H, T, S, P, shared, fusedcols):
    L, lib = _lib()
    g = torch.Generator().manual_seed(H * T + S)
    ci = 16 * H
    q = _bf(torch.randn(P, T, ci, generator=g)).to(DEV)
    nb = 1 if shared else P
    kv = _bf(torch.randn(nb, S, fusedcols * ci + (ci if fusedcols == 1 else 0), generator=g)).to(DEV)       # K | V (| other columns) side by side
    ld = kv.shape[2]
    k_off, v_off = 0, ci
    o = torch.empty(P, T, ci, dtype=torch.bfloat16, device=DEV)
    rc = lib.ovo_sam_t2i_attention(L.ptr(q), C.c_void_p(kv.data_ptr() + 2 * k_off), C.c_void_p(kv.data_ptr() + 2 * v_off), 0 if shared else S * ld, ld,
                                   L.ptr(o), P, S, T, H, 0.25, L.stream())
    L.check(rc)
    kk = kv[..., k_off:k_off + ci].float().expand(P, S, ci).reshape(P, S, H, 16).permute(0, 2, 1, 3)
    vv = kv[..., v_off:v_off + ci].float().expand(P, S, ci).reshape(P, S, H, 16).permute(0, 2, 1, 3)
    qq = q.float().reshape(P, T, H, 16).permute(0, 2, 1, 3)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) * 0.25, dim=-1) @ vv).permute(0, 2, 1, 3).reshape(P, T, ci)
    torch.testing.assert_close(o.float(), ref, atol=0.02, rtol=0.02)
    # the generic flash kernel on the same strided operands
    a = L.Attention()
    o2 = torch.empty_like(o)
    a.q, a.k, a.v, a.o = q.data_ptr(), kv.data_ptr() + 2 * k_off, kv.data_ptr() + 2 * v_off, o2.data_ptr()
    a.q_sb, a.q_sh, a.q_st = T * ci, 16, ci
    a.k_sb, a.k_sh, a.k_st = (0 if shared else S * ld), 16, ld
    a.v_sb, a.v_sh, a.v_st = (0 if shared else S * ld), 16, ld
    a.o_sb, a.o_sh, a.o_st = T * ci, 16, ci
    a.B, a.H, a.Tq, a.Tk, a.hd, a.scale = P, H, T, S, 16, 0.25
    L.check(lib.ovo_attention(C.byref(a), L.stream()))
    torch.testing.assert_close(o.float(), o2.float(), atol=0.02, rtol=0.02)


def test_t2i_attention_reports_unsupported_token_counts():
    L, lib = _lib()
    t = torch.zeros(1, 5, 128, dtype=torch.bfloat16, device=DEV)
    assert lib.ovo_sam_t2i_attention(L.ptr(t), L.ptr(t), L.ptr(t), 0, 128, L.ptr(t), 1, 5, 5, 8, 0.25, L.stream()) == L.E_UNSUPPORTED   # 8 x 5 does not divide 64


def test_i2t_attention_reads_a_column_block_of_the_fused_projection():
    """q as columns [2 ci, 3 ci) of the K | V | Q matrix (token stride 3 ci) == q as its own contiguous matrix."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(11)
    P, S, T, H = 3, 200, 8, 8
    ci = 16 * H
    fused = _bf(torch.randn(P, S, 3 * ci, generator=g)).to(DEV)
    k = _bf(torch.randn(P, T, ci, generator=g)).to(DEV)
    v = _bf(torch.randn(P, T, ci, generator=g)).to(DEV)
    qc = fused[..., 2 * ci:].contiguous()
    o1 = torch.empty(P, S, ci, dtype=torch.bfloat16, device=DEV)
    o2 = torch.empty_like(o1)
    L.check(lib.ovo_sam_i2t_attention(C.c_void_p(fused.data_ptr() + 2 * 2 * ci), S * 3 * ci, 3 * ci, L.ptr(k), L.ptr(v), L.ptr(o1), P, S, T, H, 0.25, L.stream()))
    L.check(lib.ovo_sam_i2t_attention(L.ptr(qc), S * ci, ci, L.ptr(k), L.ptr(v), L.ptr(o2), P, S, T, H, 0.25, L.stream()))
    assert torch.equal(o1, o2)
    qq = qc.float().reshape(P, S, H, 16).permute(0, 2, 1, 3)
    kk = k.float().reshape(P, T, H, 16).permute(0, 2, 1, 3)
    vv = v.float().reshape(P, T, H, 16).permute(0, 2, 1, 3)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) * 0.25, dim=-1) @ vv).permute(0, 2, 1, 3).reshape(P, S, ci)
    torch.testing.assert_close(o1.float(), ref, atol=0.02, rtol=0.02)
