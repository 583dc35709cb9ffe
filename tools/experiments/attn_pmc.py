#!/usr/bin/env python3
"""Runs the attention core kernels a few times at the benchmark shape (for rocprofv3 --pmc / --kernel-trace passes):
   python3 tools/attn_pmc.py [B] [rel_pos 0|1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rel_pos = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
gens = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 0]
plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, 16, 0)
Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
torch.manual_seed(0)
qkvh = (torch.randn(Bw, h, 3, Lp, DP, device=dev) * 0.25).to(BF)
qkvh[:, :, :, Lw:] = 0
rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev)
lse = torch.empty(Bw, h, Lp, device=dev)
ls = torch.full((h,), 2.3, device=dev)
bias = torch.randn(h, Lw, Lw, device=dev) if rel_pos else None
doh = (torch.randn(Bw, h, Lp, DP, device=dev)).to(BF)
doh[:, :, Lw:] = 0
dq = torch.empty(Bw, h, 3, Lp, DP, dtype=BF, device=dev)
dls = torch.zeros(h, device=dev)
dbias = torch.zeros(h, Lw, Lw, device=dev) if rel_pos else None
pk = ops.attn_pack_bias(bias) if rel_pos else None
for dbg in gens:
    for _ in range(4):
        a = ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr, max_chunks=32 if rel_pos else 64, bias_pack=pk)
        a.dbg = dbg
        ops.attn_fwd(a)
        a = ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, 16, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm,
                          dqkvh=dq, dlogit=dls, dbias=dbias, max_chunks=32 if rel_pos else 64, bias_pack=pk)
        a.dbg = dbg
        ops.attn_bwd(a)
torch.cuda.synchronize()
print("done")
