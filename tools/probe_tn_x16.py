#!/usr/bin/env python3
"""fc1 / qkv weight gradients with the X operand in fp32 (the residual stream as stored) vs a bf16 copy of it (GPU box)."""
import os, sys, torch, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops
from tools.perf_probe import timeit
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = 2
M, Cc, hid = B * 64800, 128, 512
x = torch.randn(M, Cc, device=dev); x16 = x.to(BF)
dh = torch.randn(M, hid, device=dev).to(BF)
lib = L.load()
for splits in (40, 64, 128):
    for tag, xx in (("fc1 X fp32", ops.op_f32(x)), ("fc1 X bf16", ops.op_bf16(x16))):
        dy = ops.op_bf16(dh)
        nb = lib.swv2_linear_wgrad_ws_bytes(M, hid, Cc, splits)
        ws = torch.zeros(nb // 4, device=dev); dW = torch.zeros(hid, Cc, device=dev)
        f = lambda: L.check(lib.swv2_linear_wgrad_ws(C.byref(dy), C.byref(xx), dW.data_ptr(), None, None, None, Cc, splits, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream), "wgrad")
        print(f"{tag} splits={splits}: {timeit(f, n=20):.1f} us")
