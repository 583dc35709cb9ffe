#!/usr/bin/env python3
"""Run a few launches of selected kernels at the benchmark shape (for rocprofv3 --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, Cc = 129600, 128
x = torch.randn(M, Cc, device=dev); xb = x.to(BF)
hb = torch.randn(M, 512, device=dev).to(BF)
dW = torch.zeros(512, Cc, device=dev); db = torch.zeros(512, device=dev)
w = ops.prep_weight(torch.randn(512, Cc, device=dev)); o = torch.empty(M, 512, dtype=BF, device=dev); o2 = torch.empty(M, 512, dtype=BF, device=dev)
b = torch.zeros(512, device=dev)
for _ in range(3):
    ops.linear_wgrad(ops.op_bf16(hb), ops.op_f32(x), dW, db, splits=64)
    ops.linear(ops.op_f32(x), w, ops.epilogue(L.EPI_BF16, o, ld=512), 512)
    ops.linear(ops.op_bf16(xb), w, ops.epilogue(L.EPI_BF16, o, ld=512), 512)
    ops.linear(ops.op_f32(x), w, ops.epilogue(L.EPI_BF16_GELU, o, ld=512, bias=b, aux_out=o2), 512)
torch.cuda.synchronize()
