#!/usr/bin/env python3
"""Per-window progress of the attention backward kernels (GPU box, diagnostic build with -DSWV2_ATTNS_STAMPS -DSWV2_ATTN1_STAMPS): s_memtime
at the end of every window of wave 8 of the first 32 workgroups of head 0, for the streamed-dQ kernel and the two-phase kernel.
usage: PROBE_B=2 tools/probe_attn_bwd_windows.py"""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = os.environ.get("PROBE_SO") or "/tmp/libswv2_wstamps.so"        # PROBE_SO: a stamped library built beforehand (tools/build_stamped.sh)
srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
if not os.environ.get("PROBE_SO"):
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_ATTNS_STAMPS", "-DSWV2_ATTN1_STAMPS",
                       "-o", so] + srcs, stderr=subprocess.DEVNULL)
L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = int(os.environ.get("PROBE_B", "2"))
plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, 16, 0)
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
lib = ctypes.CDLL(so)
nwin = Bw // (256 // h)
kernels = (("two-phase", L.ATTN_BWD_TWO_PHASE, "swv2_debug_attn1_win"), ("streamed", 0, "swv2_debug_attns_win"))
if os.environ.get("PROBE_ONLY_STREAMED"):
    kernels = kernels[1:]
outs = {}
for name, dbg, sym in kernels:
    dq, dls = torch.empty_like(qkvh), torch.zeros(h, device=dev)
    a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm, dqkvh=dq, dlogit=dls, max_chunks=256 // h)
    a.dbg = dbg
    for _ in range(5):
        ops.attn_bwd(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attn_bwd(a)
    e1.record(); torch.cuda.synchronize()
    outs[name] = (dq.float(), dls.clone())
    buf = torch.zeros(64 * 128, dtype=torch.int64)
    assert getattr(lib, sym)(ctypes.c_void_p(buf.data_ptr())) == 0
    w = buf.view(64, 128)[:32, :min(nwin, 128)].double()
    d = torch.diff(w, dim=1, prepend=torch.zeros(32, 1, dtype=torch.float64))
    m = d.mean(0)
    print(f"{name}: {e0.elapsed_time(e1) * 100:.1f} us per launch (stamped build), {nwin} windows per workgroup; ticks per window (mean over 32 workgroups):")
    print("   first 8:", " ".join(f"{x:6.0f}" for x in m[:8]), "| last 4:", " ".join(f"{x:6.0f}" for x in m[-4:]), f"| mean {m.mean():.0f}, total {w[:, -1].mean():.0f} (min {w[:, -1].min():.0f} max {w[:, -1].max():.0f})")
if len(outs) == 2:
    (a, la), (b, lb) = outs["two-phase"], outs["streamed"]
    for part, nm in enumerate(("dq", "dk", "dv")):
        x, y = a[:, :, part], b[:, :, part]
        print(f"streamed vs two-phase {nm}: max |diff| {float((x - y).abs().max()):.3e} of max {float(x.abs().max()):.3e}; equal {bool(torch.equal(x, y))}; padded rows max {float(y[:, :, Lw:].abs().max()):.1e}")
    print("d logit_scale rel diff", float(((la - lb).abs() / la.abs().clamp_min(1e-9)).max()))
