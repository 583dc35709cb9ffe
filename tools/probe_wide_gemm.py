#!/usr/bin/env python3
"""GPU box: the wide NT kernel (256 x 256 tiles) against torch on the products of the reference's embed_dim-768 model, with timing.
   python tools/probe_wide_gemm.py            (SWV2_GEMM_WIDE=0 selects the 128 x 128 kernel for comparison)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import ops, _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)
M = int(os.environ.get("M", 64800))

def timeit(fn, n=10, rounds=3):
    """min over `rounds` of the mean of n launches (the first launches of a process run at a lower clock)"""
    for _ in range(10): fn()
    best = 1e30
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

ONLY = os.environ.get("ONLY")
SHAPES = ((768, 3072, "bf16", "bf16"), (768, 3072, "f32", "bf16"), (3072, 768, "bf16", "f32"), (768, 768, "bf16", "f32"),
          (768, 2304, "f32", "f32"))
for (K, N, akind, ekind) in (SHAPES if ONLY is None else [SHAPES[int(i)] for i in ONLY.split(",")]):
    a32 = torch.randn(M, K, device=dev)
    a = a32 if akind == "f32" else a32.bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if ekind == "bf16" else torch.float32)
    op = ops.op_f32(a) if akind == "f32" else ops.op_bf16(a)
    ep = ops.epilogue(L.EPI_BF16 if ekind == "bf16" else L.EPI_F32, out, ld=N, bias=bias)
    run = lambda: ops.linear(op, w, ep, N)
    run(); torch.cuda.synchronize()
    ref = a32.bfloat16().float() @ w.float().t() + bias
    err = (out.float() - ref).abs().max().item()
    us = timeit(run)
    if os.environ.get("AB"):          # same process, same device: one tile per workgroup instead of persistent workgroups; the 128 x 128 kernel
        os.environ["SWV2_WIDE_PERSIST"] = "0"; us3 = timeit(run)
        os.environ["SWV2_WIDE_PERSIST"] = "1"; os.environ["SWV2_GEMM_WIDE"] = "0"; us4 = timeit(run)
        os.environ["SWV2_GEMM_WIDE"] = "1"
        print(f"    one tile per workgroup {us3:8.1f} us | 128 x 128 kernel {us4:8.1f} us")
    tus = timeit(lambda: torch.matmul(a32.bfloat16() if akind == "f32" else a, w.t()))
    print(f"K={K} N={N} A={akind} out={ekind}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s   max|err| {err:.3e}   "
          f"(torch bf16 matmul {tus:8.1f} us)")
