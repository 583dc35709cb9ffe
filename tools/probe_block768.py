#!/usr/bin/env python3
"""GPU box: one embed-768 block (8 heads of 96, 72 x 360 tokens) forward + backward against the bf16-emulating oracle, with the
wide kernels on / off (SWV2_GEMM_WIDE, SWV2_ATTN_WIDE): which kernel family carries which part of the deviation."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from swin_v2_weather_amd.networks import swinv2_global as N
from oracle import swin_oracle as O

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

dev = torch.device("cuda:0")
torch.manual_seed(11)
Cc, h, gh, gw, wh, ww, sh, sw, B = 768, 8, 72, 360, 9, 18, 4, 9, 1
blk = N.SwinTransformerV2CrBlock(dim=Cc, num_heads=h, feat_size=(gh, gw), window_size=(wh, ww), shift_size=(sh, sw), rel_pos=False, drop_path=0.0)
with torch.no_grad():
    for n_, p_ in blk.named_parameters():
        if n_.endswith("norm1.weight") or n_.endswith("norm2.weight"): p_.uniform_(0.5, 1.5)
        elif n_.endswith("logit_scale"): p_.copy_(torch.log(torch.tensor(10.0)) + 0.25 * torch.randn(h))
x, gy = torch.randn(B, gh, gw, Cc), torch.randn(B, gh, gw, Cc)
p = {"b." + n_: v.detach().clone().requires_grad_(True) for n_, v in blk.named_parameters()}
cfg = dict(feat_size=(gh, gw), window_size=(wh, ww), shift_size=(sh, sw), dim=Cc, num_heads=h, rel_pos=False)
from test_gpu_parity import block_cfg
xo = x.clone().requires_grad_(True)
O.set_rounding(O.bf16_round)
yo = O.block_forward(xo, p, "b.", block_cfg(gh, gw, wh, ww, sh, sw, Cc, h, False), 1, training=False)
yo.backward(gy)
O.set_rounding(None)
xe = x.clone().requires_grad_(True)
pe = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
ye = O.block_forward(xe, pe, "b.", block_cfg(gh, gw, wh, ww, sh, sw, Cc, h, False), 1, training=False)      # exact fp32
ye.backward(gy)
print(f"oracle bf16 emulation vs exact fp32: y {rel(yo, ye):.2e}  dx {rel(xo.grad, xe.grad):.2e}")
blk = blk.to(dev).eval()
for gw_, aw_ in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
    os.environ["SWV2_GEMM_WIDE"], os.environ["SWV2_ATTN_WIDE"] = gw_, aw_
    for q in blk.parameters(): q.grad = None
    xd = x.to(dev).requires_grad_(True)
    y = blk(xd)
    y.backward(gy.to(dev))
    worst = max((rel(q.grad, p["b." + n_].grad), n_) for n_, q in blk.named_parameters() if q.grad is not None and float(p["b." + n_].grad.abs().max()) > 1e-3 and not n_.endswith("logit_scale"))
    ls = rel(blk.attn.logit_scale.grad, p["b.attn.logit_scale"].grad)
    print(f"GEMM wide {gw_} attention wide {aw_}: y vs emu {rel(y, yo):.2e} (vs exact {rel(y, ye):.2e})  dx {rel(xd.grad, xo.grad):.2e}  worst dparam {worst[0]:.2e} ({worst[1]})  dlogit {ls:.2e}")
