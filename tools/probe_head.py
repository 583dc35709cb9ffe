#!/usr/bin/env python3
"""Head GEMM + loss epilogue at the benchmark shape (GPU box): timing of the plain un-patchify epilogue against the loss
epilogue, and the two backward products from the scaled residual."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops  # noqa: E402
from tools.perf_probe import timeit  # noqa: E402

dev = torch.device("cuda:0")
B, Cout, Cc, H, W = int(os.environ.get("B", 2)), 73, 128, 720, 1440
gh, gw = H // 4, W // 4
T, M, Nn = gh * gw, B * gh * gw, Cout * 16
e2d = torch.randn(M, Cc, device=dev)
wb = ops.prep_weight((torch.randn(Nn, Cc) * 0.1).to(dev))
tar = torch.randn(B, Cout, H, W, device=dev)
qw = torch.rand(H, device=dev)
y = torch.empty(B * Cout * H * W + 512, device=dev)[:B * Cout * H * W].view(B, Cout, H, W)
part = torch.empty((M + 31) // 32, 2, Cout, 2, device=dev)
RP = L.loss_resid_pitch(Nn)
resid = torch.empty(M * RP + 1024, dtype=torch.bfloat16, device=dev)[:M * RP].view(M, RP)
print("plain un-patchify:", round(timeit(lambda: ops.linear(ops.op_f32(e2d), wb, ops.epilogue(L.EPI_UNPATCH, y, p=(Cout, H, W, 0, 0)), Nn)), 1), "us")
f = lambda: ops.linear(ops.op_f32(e2d), wb, ops.epilogue(L.EPI_UNPATCH_LOSS, y, p=(Cout, H, W, 0, 0), loss=(tar, qw, part, resid, 0)), Nn)
print("loss epilogue:", round(timeit(f), 1), "us")
coef = torch.rand(B, Cout, device=dev)
wt = ops.prep_weight((torch.randn(Nn, Cc) * 0.1).to(dev), transpose=True)
de = torch.empty(M, Cc, device=dev)
dw = torch.zeros(Nn, Cc, device=dev)
print("dx from scaled residual:", round(timeit(lambda: ops.linear(ops.op_bf16_cscale(resid, coef, T, cols=Nn), wt, ops.epilogue(L.EPI_F32, de, ld=Cc), Cc)), 1), "us")
print("dW from scaled residual:", round(timeit(lambda: ops.linear_wgrad(ops.op_bf16_cscale(resid, coef, T, cols=Nn), ops.op_f32(e2d), dw, None)), 1), "us")
