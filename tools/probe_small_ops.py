#!/usr/bin/env python3
"""Which Python call sites issue the small torch kernels / copies of a training step (fills, H2D copies, elementwise)?  GPU box.
Runs a few steps of the bench workload under torch.profiler (with_stack) and prints, per aten op that launches something, the count per
step and the innermost repository frame."""
import os, sys, collections, torch
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd.networks import helpers
from swin_v2_weather_amd.utils.losses import LossHandler
from swin_v2_weather_amd.utils.optim import HipAdam
dev = torch.device("cuda:0")
H, W, B = 720, 1440, 2
p = SimpleNamespace(nettype="swin", img_size=[H, W], patch_size=4, depth=12, num_heads=8, n_in_channels=73, n_out_channels=73,
                    embed_dim=128, window_ratio=80, drop_path_rate=0.1, full_pos_embed=True, rel_pos=False, mlp_ratio=4,
                    activation_ckpt=False, residual=False, n_future=0, add_orography=False, add_landmask=False)
model = helpers.get_model(p).to(dev).train()
lp = SimpleNamespace(n_future=0, img_shape_x=H, img_shape_y=W, loss="l2", channel_weights="none", n_out_channels=73,
                     model_grid_type="equiangular")
loss_obj = LossHandler(lp).to(dev)
opt = HipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.95))
inp, tar = torch.randn(B, 73, H, W, device=dev), torch.randn(B, 73, H, W, device=dev)


def step():
    model.zero_grad()
    with loss_obj.fused_with(model, tar):
        gen = model(inp)
    loss = loss_obj(gen, tar, inp)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 3
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
cnt, dur = collections.Counter(), collections.Counter()
for e in prof.events():
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    ks = ",".join(sorted({k.name.split("(")[0][-40:] for k in e.kernels}))
    if "swv2" in e.name or any(x in ks for x in ("gemm_", "attn_", "mlp_", "proj_ln", "tn_slab", "ln_residual")):
        continue
    key = (e.name, str(e.input_shapes)[:70], ks[:60])
    cnt[key] += 1
    dur[key] += sum(k.duration for k in e.kernels)
for key, c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c / N:6.1f}/step {dur[key] / c:7.1f} us  {key[0]:24s} {key[1]:70s} {key[2]}")
