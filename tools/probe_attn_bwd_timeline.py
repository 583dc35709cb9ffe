#!/usr/bin/env python3
"""Timeline of ONE window of one workgroup of the streamed attention backward (GPU box; stamped library from tools/build_stamped.sh in PROBE_SO).
Needs the event instrumentation (swv2_debug_attns_tl) of tools/experiments/attn_bwd_stream_nobar.hip: copy that file over csrc/attn_bwd_stream.hip in a
scratch checkout (its macros select every form of LABNOTES round 6, second pass; -DSWV2_ATTNS_TAIL=0 -DSWV2_ATTNS_ISSUE_PRIO=0 -DSWV2_ATTNS_PREFETCH_ALL=0 is the
shipped kernel), build with tools/build_stamped.sh:
every wave's events in cycles since the window's start.  Tags: 15 window start, 8 + p phase-1 signal of pair p, 1 phase 1 done, 4 dK / dV stored,
0 prefetch issued, 2 pair counter reached, 3 dQ pass done, 5 commit done, 6 behind the barrier."""
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
buf = torch.zeros(16 * 24, dtype=torch.int64)
assert ctypes.CDLL(so).swv2_debug_attns_tl(ctypes.c_void_p(buf.data_ptr())) == 0
ev = buf.view(16, 24)
names = {15: "start", 20: "iss0", 21: "dma", 1: "p1done", 4: "dkdv", 0: "issued", 2: "ctr", 3: "dQ", 5: "commit", 6: "barrier"}
t0 = min(int(ev[w, 0]) & ((1 << 56) - 1) for w in range(16))
for w in range(16):
    out = []
    for k in range(24):
        v = int(ev[w, k])
        if v == 0:
            break
        tag, t = (v >> 56) & 0xff, (v & ((1 << 56) - 1)) - t0
        out.append(f"{names.get(tag, 'sig%d' % (tag - 8))}@{t}")
    print(f"wave {w:2d} (SIMD {w % 4}): " + " ".join(out))
