#!/usr/bin/env python3
"""Attention backward of the other configurations (GPU box): 16-wide heads with a CPB table (rel_pos=True) and 24-wide heads in 32-wide
slots (BASELINE configs[3]) -- time per launch.  usage: tools/probe_attn_bwd_cfg.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = int(os.environ.get("PROBE_B", "2"))
for HD, rel_pos in ((16, True), (24, False)):
    plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, HD, 0)
    Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
    torch.manual_seed(0)
    qkvh = torch.randn(Bw, h, 3, Lp, DP, device=dev)
    qkvh[..., HD:] = 0
    qkvh[:, :, :2] = torch.nn.functional.normalize(qkvh[:, :, :2], dim=-1)
    qkvh[:, :, :, Lw:] = 0
    qkvh = qkvh.to(BF).contiguous()
    oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev); lse = torch.zeros(Bw, h, Lp, device=dev)
    ls = torch.full((h,), 2.3, device=dev)
    bias = torch.randn(h, Lw, Lw, device=dev) if rel_pos else None
    pk = ops.attn_pack_bias(bias) if rel_pos else None
    nck = L.load().swv2_attn_bias_chunks(Bw) if rel_pos else 256 // h
    ops.attn_fwd(ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, HD, plan.nwh, plan.nww, plan.mask_thr, bias_pack=pk))
    doh = torch.randn(Bw, h, Lp, DP, device=dev).to(BF); doh[:, :, Lw:] = 0; doh[..., HD:] = 0
    rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
    dq, dls = torch.empty_like(qkvh), torch.zeros(h, device=dev)
    part = torch.empty(nck, h, Lw, Lw, device=dev) if rel_pos else None
    a = ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, HD, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm, dqkvh=dq, dlogit=dls,
                      dbias=None, max_chunks=nck, bias_pack=pk, dbias_ws=part)
    if rel_pos:
        a.dbias_partials = 1
    for _ in range(3):
        ops.attn_bwd(a)
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attn_bwd(a)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 50)
    print(f"attn_bwd head_dim {HD} bias {rel_pos} B={B}: " + " / ".join(f"{t:.1f}" for t in ts) + f" us per launch; checksum {float(dq.float().abs().sum()):.6e}")
