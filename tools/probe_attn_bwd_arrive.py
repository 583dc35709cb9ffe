#!/usr/bin/env python3
"""Which waves does the window's barrier of the streamed attention backward wait for?  (GPU box; PROBE_SO = the shipped kernel rebuilt with -DSWV2_ATTNS_ARRIVE:
one s_memtime per wave and window at the barrier, 8 windows of one workgroup -- no other instrumentation, so the kernel runs as in production.)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = os.environ["PROBE_SO"]
L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
plan = ops.window_plan(2, 180, 360, 9, 18, 4, 9, 8, 16, 0)
Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
torch.manual_seed(0)
qkvh = torch.randn(Bw, h, 3, Lp, DP, device=dev)
qkvh[:, :, :2] = torch.nn.functional.normalize(qkvh[:, :, :2], dim=-1)
qkvh[:, :, :, Lw:] = 0
qkvh = qkvh.to(BF).contiguous()
oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev); lse = torch.zeros(Bw, h, Lp, device=dev)
ls = torch.full((h,), 2.3, device=dev)
ops.attn_fwd(ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr))
doh = torch.randn(Bw, h, Lp, DP, device=dev).to(BF); doh[:, :, Lw:] = 0
rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
dq, dls = torch.empty_like(qkvh), torch.zeros(h, device=dev)
a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm, dqkvh=dq, dlogit=dls, max_chunks=256 // h)
for _ in range(5):
    ops.attn_bwd(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attn_bwd(a)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 50:.1f} us per launch")
buf = torch.zeros(17 * 8, dtype=torch.int64)
assert ctypes.CDLL(so).swv2_debug_attns_arrive(ctypes.c_void_p(buf.data_ptr())) == 0
t = buf.view(17, 8).double()
opened = t[16]
print("window length (barrier to barrier):", " ".join(f"{float(x):6.0f}" for x in (opened[1:] - opened[:-1])))
print("arrival at the barrier, cycles BEFORE it opens (mean over 7 windows; 0 = the wave the barrier waited for):")
for w in range(16):
    d = (opened[1:] - t[w, 1:])
    role = "phase 1" if w < 11 else f"helper {w - 11}"
    print(f"  wave {w:2d} (SIMD {w % 4}, {role:9s}): mean {float(d.mean()):6.0f}   min {float(d.min()):6.0f}   max {float(d.max()):6.0f}")
