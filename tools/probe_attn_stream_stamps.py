#!/usr/bin/env python3
"""In-kernel phase timing of the streamed-dQ attention backward (attn_bwd_stream_kernel): private build with -DSWV2_ATTNS_STAMPS; every
wave of the first workgroups of head 0 sums s_memtime deltas per phase -- GPU box, diagnostics only."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = "/tmp/libswv2_sstamps.so"
srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_ATTNS_STAMPS", "-o", so] + srcs,
                      stderr=subprocess.DEVNULL)
L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = int(os.environ.get("PROBE_B", "2"))
plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, 16, 0)
Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
qkvh = (torch.randn(Bw, h, 3, Lp, DP, device=dev) * 0.25).to(BF); qkvh[:, :, :, Lw:] = 0
oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev); lse = torch.zeros(Bw, h, Lp, device=dev)
ls = torch.full((h,), 2.3, device=dev)
a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr)
ops.attn_fwd(a)
doh = torch.randn(Bw, h, Lp, DP, device=dev).to(BF); rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
dq, dls = torch.empty_like(qkvh), torch.zeros(h, device=dev)
a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm, dqkvh=dq, dlogit=dls, max_chunks=256 // h)
for _ in range(3):
    ops.attn_bwd(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_bwd(a)
e1.record(); torch.cuda.synchronize()
buf = torch.zeros(512 * 8, dtype=torch.int64)
assert ctypes.CDLL(so).swv2_debug_attns_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
ck = torch.zeros(512 * 2, dtype=torch.int64)
assert ctypes.CDLL(so).swv2_debug_attns_clock(ctypes.c_void_p(ck.data_ptr())) == 0
ck = ck.view(512, 2).double()[:64]
print(f"window loop: {ck[:, 1].mean() / 100:.1f} us by s_memrealtime, in-kernel clock {ck[:, 0].mean() / ck[:, 1].mean() * 100:.0f} MHz, B = {B}")
NW = 16          # 11 phase-1 waves + 5 helpers
nwg = 512 // NW
pw = buf.view(512, 8).double()[:nwg * NW].view(nwg, NW, 8)
names = ["issue next window's prefetch", "phase 1 loop", "phase 2: spin on the pair counter", "phase 2: dQ + normalisation + store",
         "dK / dV normalisation + stores", "commit (prefetch wait, LDS, delta)", "the window's barrier"]
tot = pw.sum(2).mean()
print(f"attn_bwd (streamed dQ, with stamps): {e0.elapsed_time(e1) * 100:.1f} us; {nwg} workgroups, total ticks per wave mean {tot:.0f}")
for i, n in enumerate(names):
    print(f"  {n:40s} {100 * pw[:, :, i].mean() / tot:5.1f} %")
print("per wave (mean ticks): issue | phase 1 | p2 spin | p2 | dK/dV | commit | barrier")
for w in range(NW):
    m = pw[:, w, :].mean(0)
    print(f"  wave {w:2d}: " + " ".join(f"{m[i]:8.0f}" for i in range(7)))
