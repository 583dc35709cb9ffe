#!/usr/bin/env python3
"""Attention forward at the benchmark shape (GPU box): time per launch, output checksum.  usage: tools/probe_attn_fwd.py [B ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
for B in [int(x) for x in sys.argv[1:]] or [2]:
    plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, 16, 0)
    Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
    torch.manual_seed(0)
    qkvh = torch.randn(Bw, h, 3, Lp, DP, device=dev)
    qkvh[:, :, :2] = torch.nn.functional.normalize(qkvh[:, :, :2], dim=-1)
    qkvh[:, :, :, Lw:] = 0
    qkvh = qkvh.to(BF).contiguous()
    oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev); lse = torch.zeros(Bw, h, Lp, device=dev)
    ls = torch.full((h,), 2.3, device=dev)
    a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr)
    for _ in range(5):
        ops.attn_fwd(a)
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attn_fwd(a)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 50)
    try:
        import ctypes
        lib = ctypes.CDLL(L.LIB_PATH)
        buf = torch.zeros(2048 * 2, dtype=torch.int64)
        if lib.swv2_debug_fwd3_span(ctypes.c_void_p(buf.data_ptr())) == 0:
            sp = buf.view(2048, 2)[:768].double()
            items = Bw * h / 768.0
            print(f"   item loop of a workgroup: {sp[:, 0].mean():.0f} cycles (max {sp[:, 0].max():.0f}) = {sp[:, 0].mean() / items:.0f} per item, "
                  f"{sp[:, 1].mean() / 100:.1f} us, in-kernel clock {100 * sp[:, 0].mean() / sp[:, 1].mean():.0f} MHz")
    except AttributeError:
        pass
    print(f"attn_fwd B={B}: " + " / ".join(f"{t:.1f}" for t in ts) + f" us per launch; checksum {float(oh.float().sum()):.6e} lse {float(lse.sum()):.6e}")
