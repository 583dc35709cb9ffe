#!/usr/bin/env python3
"""In-kernel phase timing of the gen2 attention kernels (s_memtime stamps): builds a private copy of the library with
-DSWV2_ATTN_STAMPS (the stamps land in the padded tail of lse rows) -- GPU box, diagnostics only."""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = "/tmp/libswv2_stamps.so"
srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_ATTN_STAMPS", "-o", so] + srcs)
L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = 2
plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, 16, 0)
Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
qkvh = (torch.randn(Bw, h, 3, Lp, DP, device=dev) * 0.25).to(BF); qkvh[:, :, :, Lw:] = 0
oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev); lse = torch.zeros(Bw, h, Lp, device=dev)
ls = torch.full((h,), 2.3, device=dev)
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr)
a.dbg = dbg
for _ in range(3):
    ops.attn_fwd(a)
torch.cuda.synchronize()
nchunk = (3 * 256 + h - 1) // h
print("dbg", dbg, "workgroups per head", nchunk)
st = lse[:nchunk, :, Lw:].contiguous().view(torch.int64).view(-1, 7).cpu().double()
names = ["q operand prep", "QK mfma (11 x K=32)", "exp pass", "cvt + PV + row-sum mfma", "normalise + store", "item prologue (fragments, masks)", "barrier"]
tot = st[:, :7].sum(1)
print(f"forward fwd3, wave 0 of {len(st)} workgroups; total ticks/wave mean {tot.mean():.0f} (s_memtime ticks)")
for i, n in enumerate(names):
    print(f"  {n:30s} {st[:, i].mean():10.0f}  {100 * st[:, i].mean() / tot.mean():5.1f} %   (min {st[:, i].min():.0f} max {st[:, i].max():.0f})")
