"""Diagnosis of the non-deterministic two-workgroups-per-CU variant of k_mlp_stream (mlp_stream.hip).  Needs a debug build:
    python -m ovo_amd.build --force --gemm-debug && python tools/mlp_race.py
Variant 2 (two 256-thread workgroups per CU) at 786 432 rows fails in every launch; each run below changes ONE thing (OVO_MLP_DBG bits, see MlpArgs) and
reports launches that differ from the production variant's output, the weights found WRONG IN LDS by the in-kernel check (pieces another wave brought in,
compared with their global source right after the barrier / after the chunk's products), and the LDS base of the workgroups that saw them."""
import os, sys
os.environ["OVO_KNOBS_DYNAMIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
DEV = "cuda:0"
lib = L.load()
rows, d, k1 = int(os.environ.get("ROWS", 786432)), 112, 128
hid = 4 * d
g = torch.Generator().manual_seed(rows + d)
x0 = (torch.randn(rows, d, generator=g) * 2 + 0.5).to(DEV)
gamma, beta = (torch.randn(d, generator=g) * 0.5 + 1).to(DEV), (torch.randn(d, generator=g) * 0.1).to(DEV)
w1 = torch.zeros(hid, k1, dtype=torch.bfloat16, device=DEV)
w1[:, :d] = (torch.randn(hid, d, generator=g) * d ** -0.5).to(DEV, torch.bfloat16)
w2 = (torch.randn(d, hid, generator=g) * hid ** -0.5).to(DEV, torch.bfloat16)
b1, b2 = torch.randn(hid, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
NB = (rows + 15) // 16
HW = 5 + hid // 64
out = torch.zeros(8 + 4 * 200 + HW * NB + 64 * NB, dtype=torch.int32, device=DEV)
os.environ["OVO_MLP_DBG_OUT"] = str(out.data_ptr())
def call(xf):
    rc = lib.ovo_mlp_f32(xf.data_ptr(), rows, d, gamma.data_ptr(), beta.data_ptr(), 1e-6, w1.data_ptr(), k1, b1.data_ptr(), hid, w2.data_ptr(), hid, b2.data_ptr(), L.stream())
    assert rc == 0, rc
os.environ["OVO_MLP_RB"], os.environ["OVO_MLP_DBG"] = "1", "0"
ref = x0.clone(); call(ref); torch.cuda.synchronize()
def run(variant, dbg, what, iters=12):
    os.environ["OVO_MLP_RB"], os.environ["OVO_MLP_DBG"] = str(variant), str(dbg)
    out.zero_()
    bad, blocks = 0, 0
    for it in range(iters):
        xf = x0.clone(); call(xf); torch.cuda.synchronize()
        if not torch.equal(xf, ref):
            bad += 1
            blocks += int(((xf - ref).abs().amax(1) > 0).view(-1, 16).any(1).sum())
    h = out.cpu().tolist()
    print(f"variant {variant} dbg {dbg:3d} {what:78s}: {bad:2d} of {iters} launches differ ({blocks} row blocks); LDS check: {h[0]} wrong pieces, {h[1]} = the buffer's "
          f"previous chunk; workgroup launches with a non-zero LDS base {h[2]}")
    recs = [(h[8 + 4 * i], h[9 + 4 * i] >> 16, h[9 + 4 * i] & 0xffff, h[10 + 4 * i] & 0xff, (h[10 + 4 * i] >> 12) & 0x1ff, h[11 + 4 * i]) for i in range(min(h[3], 200))]
    if recs:
        print("      first records (workgroup, chunk, piece, LDS base granules, LDS size granules, 0 = after the barrier / 1 = after the products):", recs[:10])
        print("      workgroups < 256:", sum(r[0] < 256 for r in recs), " >= 256:", sum(r[0] >= 256 for r in recs), " base == 0:", sum(r[3] == 0 for r in recs), " pieces < P1 (W1):",
              sum(r[2] < 1024 for r in recs), " chunks:", sorted(set(r[1] for r in recs)))
    sys.stdout.flush()
def stages(variant, dbg, iters=8):
    """Per row block: hashes of the LayerNorm-ed fragments, every chunk's hidden fragments and the FC2 accumulators (dbg & 256), against variant 1's."""
    os.environ["OVO_MLP_RB"], os.environ["OVO_MLP_DBG"] = "1", "256"
    out.zero_(); xf = x0.clone(); call(xf); torch.cuda.synchronize()
    want = out[808:808 + HW * NB].view(NB, HW).clone()
    assert torch.equal(xf, ref)
    os.environ["OVO_MLP_RB"], os.environ["OVO_MLP_DBG"] = str(variant), str(dbg | 256)
    names = ["x as loaded", "mean / rstd", "gamma / beta as read from LDS", "LN fragments"] + [f"hidden chunk {c}" for c in range(HW - 5)] + ["FC2 accumulators"]
    first = {}
    n_bad_out, n_hash_only = 0, 0
    for it in range(iters):
        out.zero_(); xf = x0.clone(); call(xf); torch.cuda.synchronize()
        got = out[808:808 + HW * NB].view(NB, HW)
        bad_out = ((xf - ref).abs().amax(1) > 0).view(-1, 16).any(1)
        bad_hash = (got != want)
        for b in torch.nonzero(bad_out | bad_hash.any(1)).flatten().tolist():
            st = torch.nonzero(bad_hash[b]).flatten().tolist()
            key = names[st[0]] if st else "none (only the stored rows differ)"
            first[key] = first.get(key, 0) + 1
            n_bad_out += int(bad_out[b]); n_hash_only += int(not bad_out[b])
            if sum(first.values()) <= 6:
                print(f"      launch {it} row block {b} (workgroup slot {(b // (4 * 2)) % 512 if variant == 2 else -1}): output wrong {bool(bad_out[b])}; stages that differ: {[names[k] for k in st][:5]}")
    print(f"variant {variant} dbg {dbg}: wrong row blocks {n_bad_out} (+{n_hash_only} with a wrong hash but right output) in {iters} launches; FIRST wrong stage: {first}")
    sys.stdout.flush()
def again(variant, iters=10):
    """dbg & 512: the row sum recomputed from the same registers right after it was used."""
    import struct
    os.environ["OVO_MLP_RB"], os.environ["OVO_MLP_DBG"] = str(variant), "512"
    out.zero_()
    bad = 0
    for it in range(iters):
        xf = x0.clone(); call(xf); torch.cuda.synchronize()
        bad += int(not torch.equal(xf, ref))
    h = out.cpu().tolist()
    f = lambda u: struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
    print(f"variant {variant} dbg 512: {bad} of {iters} launches differ; lanes whose row sum / squared deviations / rstd do not repeat: {h[5]}")
    for i in range(min(h[5], 40)):
        r = h[8 + 800 - 320 + 8 * i: 8 + 800 - 320 + 8 * i + 8]
        r2 = h[8 + 800 - 640 + 8 * i: 8 + 800 - 640 + 8 * i + 8]
        if i < 12 or (r[2] & 15) == 0:
            print(f"      workgroup {r[0]} row block {r[1]} lane {r[2]} (row {r[2] & 15}, quarter {r[2] >> 4}): sum as used {f(r[3])!r} again {f(r[4])!r}; squared deviations as used {f(r[5])!r} again "
                  f"{f(r[6])!r}; this lane's part as used {f(r2[0])!r} again {f(r[7])!r}; mean {f(r2[1])!r} again {f(r2[2])!r}; exec at use {(r2[4] & 0xffffffff) << 32 | (r2[3] & 0xffffffff):016x} now {(r2[6] & 0xffffffff) << 32 | (r2[5] & 0xffffffff):016x}")
    sys.stdout.flush()
def values(variant, iters=10):
    """dbg & 1024: (sum, squared deviations, mean, rstd) of every row, against variant 1's and against torch's on the same x."""
    os.environ["OVO_MLP_RB"], os.environ["OVO_MLP_DBG"] = "1", "1024"
    out.zero_(); xf = x0.clone(); call(xf); torch.cuda.synchronize()
    base = 808 + HW * NB
    want = out[base:base + 64 * NB].view(torch.float32).view(NB * 16, 4).clone()
    os.environ["OVO_MLP_RB"] = str(variant)
    shown = 0
    for it in range(iters):
        out.zero_(); xf = x0.clone(); call(xf); torch.cuda.synchronize()
        got = out[base:base + 64 * NB].view(torch.float32).view(NB * 16, 4)
        bad_rows = torch.nonzero((got.view(torch.int32) != want.view(torch.int32)).any(1)).flatten()[:rows]
        wrong_out = torch.nonzero(((xf - ref).abs().amax(1) > 0)).flatten()
        print(f"variant {variant} launch {it}: rows with different statistics {bad_rows.numel()}, rows with a wrong result {wrong_out.numel()}")
        for r in bad_rows.tolist()[:6]:
            if shown < 24:
                shown += 1
                xs = x0[r].double()
                print(f"      row {r} (block {r // 16}): sum, sq.dev, mean, rstd = {got[r].tolist()}  variant 1: {want[r].tolist()}  torch: sum {float(xs.sum()):.6f} "
                      f"sq.dev {float(((xs - xs.mean()) ** 2).sum()):.6f}")
    sys.stdout.flush()
if os.environ.get("STAGES") == "again":
    again(2, 20); again(2, 20); again(1)
    sys.exit(0)
if os.environ.get("STAGES", "1") != "0":
    values(2); again(2); again(1)
    stages(2, 0); stages(2, 128); stages(3, 0)
if os.environ.get("STAGES") == "only":
    sys.exit(0)
run(1, 0, "production: ONE 512-thread workgroup per CU")
run(1, 4 + 8, "  + LDS check")
run(2, 0, "TWO 256-thread workgroups per CU")
run(2, 4, "  + LDS check after the barrier")
run(2, 8, "  + LDS check after the products")
run(2, 64, "  padded to one workgroup per CU")
run(2, 64 + 4 + 8, "  padded + both LDS checks")
run(2, 16, "  + ~2000 idle cycles between the barrier and the first fragment read")
run(2, 16 + 4, "  + idle cycles + LDS check after them")
run(2, 32, "  + every wave reads its last piece back before it enters the barrier")
run(2, 32 + 4, "  + read-back + LDS check")
run(2, 128, "  next chunk's DMA issued AFTER this chunk's products")
run(2, 128 + 4 + 8, "  late DMA + both LDS checks")
run(2, 2, "  wait for every DMA right after its issue")
run(2, 2 + 4 + 8, "  wait at issue + both LDS checks")
run(2, 1, "  + a barrier after every chunk's products")
run(3, 0, "variant 3 (4 row blocks per wave)")
run(3, 4 + 8, "variant 3 + both LDS checks")
