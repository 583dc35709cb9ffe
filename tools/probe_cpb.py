#!/usr/bin/env python3
"""Timing probe of the continuous-position-bias kernels at the benchmark window (9x18, 8 heads, hidden 384); GPU box."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import ops
dev = torch.device('cuda:0')
wh, ww, h, Hd = 9, 18, 8, 384
L2 = (wh * ww) ** 2
w1, b1, w2, b2 = torch.randn(Hd, 2, device=dev), torch.randn(Hd, device=dev), torch.randn(h, Hd, device=dev), torch.randn(h, device=dev)
keep = torch.nn.functional.dropout(torch.ones(L2, Hd, dtype=torch.bfloat16, device=dev), 0.125, True)
bias = torch.empty(h, wh * ww, wh * ww, device=dev)
g = [torch.zeros_like(t) for t in (w1, b1, w2, b2)]
dbias = torch.randn_like(bias)
def timeit(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for k in (keep, None):
    print("mask" if k is not None else "eval", "cpb_fwd %.1f us" % timeit(lambda: ops.cpb_fwd(w1, b1, w2, b2, k, bias, wh, ww, h, Hd, 0.125)),
          "cpb_bwd %.1f us" % timeit(lambda: ops.cpb_bwd(dbias, w1, b1, w2, k, *g, wh, ww, h, Hd, 0.125)))
print("mask draw %.1f us" % timeit(lambda: torch.nn.functional.dropout(torch.ones(L2, Hd, dtype=torch.bfloat16, device=dev), 0.125, True)))
