#!/usr/bin/env python3
"""Attention backward at the benchmark shape (GPU box): the streamed-dQ kernel (csrc/attn_bwd_stream.hip) against the two-phase kernel
(csrc/attn.hip, dbg = SWV2_ATTN_BWD_TWO_PHASE) -- bit-equality of d(qkv), times of both, optionally the forward.
usage: tools/probe_attn_bwd.py [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
Bs = [int(x) for x in sys.argv[1:]] or [2]
shifted = os.environ.get("PROBE_SHIFT", "1") != "0"
for B in Bs:
    plan = ops.window_plan(B, 180, 360, 9, 18, 4 if shifted else 0, 9 if shifted else 0, 8, 16, 0)
    Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
    torch.manual_seed(0)
    qkvh = torch.randn(Bw, h, 3, Lp, DP, device=dev)
    qkvh[:, :, :2] = torch.nn.functional.normalize(qkvh[:, :, :2], dim=-1)
    qkvh[:, :, :, Lw:] = 0
    qkvh = qkvh.to(BF).contiguous()
    oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev); lse = torch.zeros(Bw, h, Lp, device=dev)
    ls = torch.full((h,), 2.3, device=dev)
    a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr)
    ops.attn_fwd(a)
    doh = torch.randn(Bw, h, Lp, DP, device=dev).to(BF); doh[:, :, Lw:] = 0
    rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
    out = {}
    for name, dbg in (("two-phase", L.ATTN_BWD_TWO_PHASE), ("streamed", 0)):
        dq, dls = torch.full_like(qkvh, float("nan")), torch.zeros(h, device=dev)
        a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm, dqkvh=dq, dlogit=dls,
                          max_chunks=256 // h)
        a.dbg = dbg
        ops.attn_bwd(a)
        torch.cuda.synchronize()
        out[name] = (dq.clone(), dls.clone())
        for _ in range(3):
            ops.attn_bwd(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attn_bwd(a)
        e1.record(); torch.cuda.synchronize()
        print(f"attn_bwd {name:10s} B={B}: {e0.elapsed_time(e1) * 50:.1f} us per launch")
    d0, d1 = out["two-phase"], out["streamed"]
    nan = int(torch.isnan(d1[0][:, :, :, :Lw].float()).sum())
    same = torch.equal(d0[0][:, :, :, :Lw], d1[0][:, :, :, :Lw])
    diff = float((d0[0][:, :, :, :Lw].float() - d1[0][:, :, :, :Lw].float()).abs().max())
    print(f"d(qkv) bit-identical: {same} (max abs diff {diff:.3e}, NaNs {nan}); d logit_scale rel diff {float((d0[1] - d1[1]).abs().max() / d0[1].abs().max()):.2e}")
    for part, nm in enumerate(("dq", "dk", "dv")):
        x, y = d0[0][:, :, part, :Lw].float(), d1[0][:, :, part, :Lw].float()
        print(f"  {nm}: max abs diff {float((x - y).abs().max()):.3e}  (|ref| max {float(x.abs().max()):.3e})")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a = ops.attn_args(qkvh, ls, None, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr)
    for _ in range(3):
        ops.attn_fwd(a)
    e0.record()
    for _ in range(20):
        ops.attn_fwd(a)
    e1.record(); torch.cuda.synchronize()
    print(f"attn_fwd B={B}: {e0.elapsed_time(e1) * 50:.1f} us per launch")
