#!/usr/bin/env python3
"""swv2_block_wgrad at the benchmark block shape (C 128, hidden 512, 8 heads x 16, 176-row windows): the slab kernel
(gemm_tn_slab.hip, slices = 0) against the 128 x 128 tile kernel (explicit slice count) and against torch on the same bf16
operands; event-timed.  GPU box, diagnostics.  Usage: tools/probe_wgrad_slab.py [local batch] [windows per sample]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 400
Cc, hid, h, DP, Lw, Lp = (192, 768, 8, 32, 162, 176) if os.environ.get("PROBE_SET") == "1" else (128, 512, 8, 16, 162, 176)
HD = h * DP
T = NW * Lw
M, Bw = B * T, B * NW
Mw = Bw * Lp
lib = L.load()
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
da2 = rn(M, Cc).to(BF); hpre = rn(M, hid).to(BF); dh = rn(M, hid).to(BF); x1 = rn(M, Cc); x = rn(M, Cc)
da1 = rn(Mw, Cc).to(BF)
oh = rn(Bw, h, 1, Lp, DP).to(BF); dqkv = rn(Bw, h, 3, Lp, DP).to(BF)
# window table: row (bw, t) -> a permutation of the image rows; padded rows (t >= L) -> -1 with zero dY rows
perm = torch.randperm(M, device=dev, generator=g).to(torch.int32).view(Bw, Lw)
rowidx = torch.full((Bw, Lp), -1, dtype=torch.int32, device=dev)
rowidx[:, :Lw] = perm
rowidx = rowidx.view(-1).contiguous()
dqkv[:, :, :, Lw:, :] = 0
da1.view(Bw, Lp, Cc)[:, Lw:, :] = 0
ident128 = torch.arange(HD, dtype=torch.int32, device=dev)
qkv_map = torch.arange(3 * HD, dtype=torch.int32, device=dev)


def items(outs):
    it = (L.WgradItem * 4)()
    ops_ = [(ops.op_bf16(da2), ops.op_bf16(hpre, gelu=True), None, None, hid),
            (ops.op_bf16(dh), ops.op_f32(x1), None, None, Cc),
            (ops.op_bf16(da1), ops.op_heads(oh, Bw, h, 1, Lp, DP), None, ident128, HD),
            (ops.op_heads(dqkv, Bw, h, 3, Lp, DP), ops.op_f32(x, rows=Mw, rowidx=rowidx), qkv_map, None, Cc)]
    keep = []
    for i, (dy, xx, nmap, kmap, ldw) in enumerate(ops_):
        it[i].dy, it[i].x = dy, xx
        it[i].dW, it[i].db = outs[i][0].data_ptr(), outs[i][1].data_ptr()
        it[i].nmap = None if nmap is None else nmap.data_ptr()
        it[i].kmap = None if kmap is None else kmap.data_ptr()
        it[i].ldw = ldw
        keep.append((dy, xx))
    return it, keep


def outs():
    return [(torch.zeros(Cc, hid, device=dev), torch.zeros(Cc, device=dev)), (torch.zeros(hid, Cc, device=dev), torch.zeros(hid, device=dev)),
            (torch.zeros(Cc, HD, device=dev), torch.zeros(Cc, device=dev)), (torch.zeros(3 * HD, Cc, device=dev), torch.zeros(3 * HD, device=dev))]


nb = lib.swv2_block_wgrad_ws_bytes(Cc, hid, h * DP, 0)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream


def run(slices, o):
    it, keep = items(o)
    L.check(lib.swv2_block_wgrad(it, slices, ws.data_ptr(), nb, st()), "swv2_block_wgrad")
    return keep


def timeit(slices, n=20):
    o = outs()
    it, keep = items(o)
    f = lambda: L.check(lib.swv2_block_wgrad(it, slices, ws.data_ptr(), nb, st()), "swv2_block_wgrad")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


new, old = outs(), outs()
OLD_SL = 40 if Cc == 128 else 8
run(0, new); run(OLD_SL, old)
torch.cuda.synchronize()
# torch reference on the same bf16 operands (fp32 accumulate)
f32 = torch.float32
gel = torch.nn.functional.gelu(hpre.float()).to(BF).float()
xg = torch.zeros(Mw, Cc, device=dev); ok = rowidx >= 0; xg[ok] = x[rowidx[ok].long()].to(BF).float()
ohm = oh[:, :, 0].permute(0, 2, 1, 3).reshape(Mw, h * DP).float()
dq = dqkv.permute(0, 3, 2, 1, 4).reshape(Mw, 3 * h * DP).float()
ref = [(da2.float().t() @ gel, da2.float().sum(0)), (dh.float().t() @ x1.to(BF).float(), dh.float().sum(0)),
       (da1.float().t() @ ohm, da1.float().sum(0)), (dq.t() @ xg, dq.sum(0))]
names = ["fc2", "fc1", "proj", "qkv"]
bad = False
for i, nm in enumerate(names):
    for k, what in enumerate(("dW", "db")):
        r, a, b = ref[i][k], new[i][k], old[i][k]
        sc = float(r.abs().max())
        en, eo = float((a - r).abs().max()) / sc, float((b - r).abs().max()) / sc
        print(f"{nm:5s} {what}: slab vs torch {en:.2e}   tile kernel vs torch {eo:.2e}   slab vs tile {float((a - b).abs().max()) / sc:.2e}")
        bad |= not (en < 2e-3)
if os.environ.get("SWV2_SLAB_STAMPS"):        # library built with -DSWV2_SLAB_STAMPS (tools/build_variant.sh gemm_tn_slab.hip stamps -DSWV2_SLAB_STAMPS)
    wg = [int(v) for v in os.environ["SWV2_SLAB_STAMPS"].split(",")]
    nwg = sum(wg)
    st_ = ws[nb - nwg * 64:].view(torch.int64).view(nwg, 8).cpu().double()
    names_ = ["wait for DMA", "barrier", "issue DMA", "zero / convert (+ barriers)", "fragment reads (+ GELU)", "MFMA", "epilogue", "prologue"]
    o = 0
    for i, nm in enumerate(names):
        blk = st_[o:o + wg[i]]; o += wg[i]
        tot = blk.sum(1)
        print(f"{nm}: {wg[i]} workgroups, wave 0 lifetime mean {tot.mean() / 100:.1f} us (min {tot.min() / 100:.1f}, max {tot.max() / 100:.1f})")
        for k, n_ in enumerate(names_):
            print(f"    {n_:30s} {blk[:, k].mean() / 100:8.2f} us {100 * blk[:, k].mean() / tot.mean():5.1f} %")
    sys.exit(0)
new2 = outs()
run(0, new2)
torch.cuda.synchronize()
det = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(new, new2))
print("deterministic:", det)
t_new, t_old = timeit(0), timeit(OLD_SL)
mb = (M * (2 * Cc + 2 * hid + 2 * hid + 4 * Cc) + Mw * (2 * Cc + 2 * HD + 6 * HD) + M * 4 * Cc) / 1e6
print(f"B={B}: slab {t_new:.1f} us, tile kernel {t_old:.1f} us (both incl. the reduction launch); operand bytes {mb:.0f} MB -> "
      f"{mb / t_new:.2f} TB/s vs {mb / t_old:.2f} TB/s")
sys.exit(1 if (bad or not det) else 0)
