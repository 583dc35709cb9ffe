#!/usr/bin/env python3
"""Ablation probe of mlp_fwd: times the kernel as built (`normal`) or privately rebuilt with a macro given as the first
argument (SWV2_MLP_GELU_ABL=1: no GELU, wrong results; SWV2_MLP_GELU_ABL=2: erf formula instead of the table) -- GPU box."""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
variant = sys.argv[1] if len(sys.argv) > 1 else "normal"
if variant != "normal":                       # e.g. SWV2_MLP_GELU_ABL=1 (no GELU), SWV2_MLP_GELU_ABL=2 (formula instead of the table)
    so = "/tmp/libswv2_probe.so"
    srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-D" + variant, "-o", so] + srcs,
                          stderr=subprocess.DEVNULL)
    L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0")
T, Cc, hid = 64800, 128, 512
M = 2 * T
w1, w2 = ops.prep_weight(torch.randn(hid, Cc, device=dev) * 0.1), ops.prep_weight(torch.randn(Cc, hid, device=dev) * 0.1)
z = lambda n: torch.zeros(n, device=dev)
xs = [torch.randn(M, Cc, device=dev) for _ in range(6)]          # rotate inputs: no help from the infinity cache
g, b1, b2, bt = torch.ones(Cc, device=dev), z(hid), z(Cc), z(Cc)
def run(i):
    return ops.mlp_fwd(xs[i % len(xs)], w1, b1, w2, b2, g, bt, None, T)
for i in range(6):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(30):
    run(i)
e1.record(); torch.cuda.synchronize()
print(f"mlp_fwd {variant}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per launch (incl. output allocation)")
