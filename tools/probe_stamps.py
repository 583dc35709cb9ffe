#!/usr/bin/env python3
"""In-kernel phase timing of mlp_fwd_kernel (s_memtime stamps): builds a private copy of the library with
-DSWV2_MLP_STAMPS (the stamps overwrite the `mean` output) -- GPU box, diagnostics only."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = "/tmp/libswv2_stamps.so"
srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_MLP_STAMPS", "-o", so] + srcs)
L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0")
T, Cc, hid = 64800, 128, 512
M = 2 * T
x = torch.randn(M, Cc, device=dev)
w1, w2 = ops.prep_weight(torch.randn(hid, Cc, device=dev) * 0.1), ops.prep_weight(torch.randn(Cc, hid, device=dev) * 0.1)
z = lambda n: torch.zeros(n, device=dev)
for _ in range(3):
    y, hpre, a2, mean, rstd = ops.mlp_fwd(x, w1, z(hid), w2, z(Cc), torch.ones(Cc, device=dev), z(Cc), None, T)
torch.cuda.synchronize()
rows_wg = 128                                             # MT = 2 at C = 128
st = torch.cat([mean[b * rows_wg: b * rows_wg + 64].view(torch.int64).view(4, 8) for b in range(0, (M + rows_wg - 1) // rows_wg, 97)]).cpu().double()
names = ["prologue", "p1 mfma issue", "gelu", "p2 mfma", "commit", "stores", "barrier", "epilogue"]
print("wave-avg ticks (100 MHz s_memtime => x21 shader cycles at 2.1 GHz):")
for i, n in enumerate(names):
    print(f"  {n:14s} {st[:, i].mean():10.0f}  (min {st[:, i].min():.0f} max {st[:, i].max():.0f})")
print("  total", st.sum(1).mean())

# ---- backward
dy = torch.randn(M, Cc, device=dev)
dg, db = z(Cc), z(Cc)
w2t, w1t = ops.prep_weight(torch.randn(Cc, hid, device=dev) * 0.1, transpose=True), ops.prep_weight(torch.randn(hid, Cc, device=dev) * 0.1, transpose=True)
y, hpre, a2, mean, rstd = ops.mlp_fwd(x, w1, z(hid), w2, z(Cc), torch.ones(Cc, device=dev), z(Cc), None, T)
mean.zero_(); rstd.fill_(1.0)
import swin_v2_weather_amd.ops as O_
da2 = torch.empty(M, Cc, dtype=torch.bfloat16, device=dev); dh = torch.empty(M, hid, dtype=torch.bfloat16, device=dev); dx = torch.empty(M, Cc, device=dev)
ws = torch.zeros(L.load().swv2_mlp_bwd_ws_floats(M, Cc) + 65536, device=dev)
b = L.MlpBwdArgs()
b.dy, b.a2, b.mean, b.rstd, b.gamma, b.hpre, b.w2t, b.w1t = (t_.data_ptr() for t_ in (dy, a2, mean, rstd, torch.ones(Cc, device=dev), hpre, w2t, w1t))
b.da2, b.dh, b.dx, b.dgamma, b.dbeta, b.ws = (t_.data_ptr() for t_ in (da2, dh, dx, dg, db, ws))
b.M, b.C, b.hidden, b.rows_per_sample = M, Cc, hid, T
for _ in range(3):
    L.load().swv2_mlp_bwd(ctypes.byref(b), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
nb = (M + 127) // 128                                # workgroups (MT = 2: 128 rows each)
o0 = nb * 2 * Cc // 2
st = ws.view(torch.int64)[o0: o0 + ((nb + 96) // 97) * 4 * 8].view(-1, 8).cpu().double()
st = st[st.sum(1) > 0]
names = ["LN prologue", "col sums+commit0", "p1 mfma", "gelu'+p2", "commit", "stores", "barrier", "epilogue"]
print("backward, per wave:")
for i, n in enumerate(names):
    print(f"  {n:16s} {st[:, i].mean():10.0f}  (min {st[:, i].min():.0f} max {st[:, i].max():.0f})")
print("  total", st.sum(1).mean(), "waves sampled", len(st))
