#!/usr/bin/env python3
"""Timing probe of the stage-level CPB kernels at the benchmark geometry (9x18 window, 8 heads, hidden 384, depth 12); GPU box.
Variants separate the parts: nchunk = 32 / 1 (staging of the attention workgroups' tables), with / without the keep bits."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import ops
dev = torch.device('cuda:0')
wh, ww, h, Hd, nblk = 9, 18, 8, 384, 12
Lw = wh * ww; L2 = Lw * Lw
ps = [[torch.randn(Hd, 2, device=dev), torch.randn(Hd, device=dev), torch.randn(h, Hd, device=dev), torch.randn(h, device=dev)] for _ in range(nblk)]
ptab = torch.tensor([t.data_ptr() for p in ps for t in p], dtype=torch.int64).to(dev)
bits = torch.empty(nblk, L2, Hd // 8, dtype=torch.int32, device=dev).random_()
bias = torch.empty(nblk, h, Lw, Lw, device=dev)
n = 3 * Hd + h * Hd + h
grads = torch.zeros(nblk, n, device=dev)
def timeit(fn, n_=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n_): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n_ * 1e3
for kb, tag in ((bits, "bits"), (None, "eval")):
    print(tag, "fwd_multi %.1f us" % timeit(lambda: ops.cpb_fwd_multi(ptab, nblk, kb, bias, wh, ww, h, Hd, 0.125)))
    for nch in (32, 1):
        dt = torch.randn(nblk, nch, h, Lw, Lw, device=dev)
        print(tag, f"bwd_multi nchunk={nch}: %.1f us (incl. fold)" % timeit(lambda: ops.cpb_bwd_multi(dt, nch, ptab, nblk, kb, grads, wh, ww, h, Hd, 0.125)))
        del dt
print("pack_multi %.1f us" % timeit(lambda: ops.attn_pack_bias_multi(bias)))
print("bits draw %.1f us" % timeit(lambda: torch.empty(nblk, L2, Hd // 8, dtype=torch.int32, device=dev).random_()))
