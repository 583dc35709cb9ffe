#!/usr/bin/env python3
"""Fused MLP kernels at the benchmark shape (GPU box): forward with / without the saved pre-activation, backward reading it
back / rebuilding it from x."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd import ops  # noqa: E402
from tools.perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 2))
M, Cc, hid, T = B * 64800, 128, 512, 64800
x = torch.randn(M, Cc, device=dev)
w1, b1 = (torch.randn(hid, Cc, device=dev) * 0.1), torch.randn(hid, device=dev) * 0.1
w2, b2 = (torch.randn(Cc, hid, device=dev) * 0.1), torch.randn(Cc, device=dev) * 0.1
gm, bt = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
sc = torch.ones(B, device=dev)
w1b, w2b, w1t, w2t = ops.prep_weight(w1), ops.prep_weight(w2), ops.prep_weight(w1, transpose=True), ops.prep_weight(w2, transpose=True)
for keep in (True, False):
    print(f"mlp_fwd keep_hpre={keep}:", round(timeit(lambda: ops.mlp_fwd(x, w1b, b1, w2b, b2, gm, bt, sc, T, keep_hpre=keep), n=20), 1), "us")
y, hpre, a2, mean, rstd = ops.mlp_fwd(x, w1b, b1, w2b, b2, gm, bt, sc, T)
dy = torch.randn(M, Cc, device=dev)
dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
print("mlp_bwd reading hpre:", round(timeit(lambda: ops.mlp_bwd(dy, a2, mean, rstd, gm, sc, hpre, w2t, w1t, dg, db, T), n=20), 1), "us")
print("mlp_bwd recompute   :", round(timeit(lambda: ops.mlp_bwd(dy, a2, mean, rstd, gm, sc, None, w2t, None, dg, db, T, x=x, w1=w1b, b1=b1), n=20), 1), "us")
