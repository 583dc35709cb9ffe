#!/usr/bin/env python3
"""In-kernel timing of the wave-per-head attention backward (csrc/attn3.hip): private build with -DSWV2_ATTN3_STAMPS; the
waves of head 0 leave s_memtime (100 MHz) at wave start / key-tile loop start / loop end / end -- GPU box, diagnostics only."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = "/tmp/libswv2_a3stamps.so"
srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
if not os.environ.get("SWV2_PROBE_NOBUILD"):          # under rocprofv3 the library is built beforehand (no child processes there)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_ATTN3_STAMPS", "-DSWV2_A3_ABL=" + os.environ.get("A3_ABL", "0"), "-o", so] + srcs,
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
doh = torch.randn(Bw, h, Lp, DP, device=dev).to(BF); doh[:, :, Lw:] = 0
rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
dq, dls = torch.empty_like(qkvh), torch.zeros(h, device=dev)
a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm, dqkvh=dq, dlogit=dls)
a.dbg = 256 | (512 if os.environ.get("A3_NOSPLIT") else 0)       # the wave-per-head kernel is opt-in
for _ in range(3):
    ops.attn_bwd(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_bwd(a)
e1.record(); torch.cuda.synchronize()
buf = torch.zeros(1024 * 8, dtype=torch.int64)
assert ctypes.CDLL(so).swv2_debug_attn3_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
st = buf.view(1024, 8)[:min(1024, Bw * h)].double()
print(f"attn_bwd3: {e0.elapsed_time(e1) * 100:.1f} us per launch; per-wave cycle sums over its units, first {len(st)} waves")
for i, n in enumerate(["prologue (loads, delta, operands)", "per-key-tile set-up", "query-tile steps", "dK / dV finalisation + stores",
                       "dQ epilogue + stores", "wave lifetime"]):
    print(f"  {n:36s} mean {st[:, i].mean():9.0f}  min {st[:, i].min():9.0f}  max {st[:, i].max():9.0f}")
