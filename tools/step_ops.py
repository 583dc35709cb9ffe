#!/usr/bin/env python3
"""torch.profiler view of one benchmark step: which aten ops launch the non-swv2 kernels (GPU box).
STEP_OPS_RELPOS=1: the CPB-bias configuration; STEP_OPS_DDP=1: under a one-rank RCCL group."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from types import SimpleNamespace
from torch.profiler import profile, ProfilerActivity
from swin_v2_weather_amd.networks.helpers import get_model
from swin_v2_weather_amd.utils.losses import LossHandler

dev = torch.device("cuda:0")
import argparse
ns = SimpleNamespace(depth=12, embed_dim=128, heads=8, height=720, width=1440, window_ratio=80, rel_pos=int(os.environ.get("STEP_OPS_RELPOS", "0")), drop_path_rate=0.1)
p = bench.model_params(ns)
model = get_model(p).to(dev).train()
lp = SimpleNamespace(n_future=0, img_shape_x=720, img_shape_y=1440, loss="l2", channel_weights="none", n_out_channels=73, model_grid_type="equiangular")
loss_obj = LossHandler(lp).to(dev)
from swin_v2_weather_amd.utils.optim import HipAdam
opt = HipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.95))
net = model
if os.environ.get("STEP_OPS_DDP", "0") == "1":          # one-rank RCCL group: what DDP adds to the step
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from swin_v2_weather_amd.networks.helpers import enable_ddp_bucket_grads
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=12)
    enable_ddp_bucket_grads(net)
x, y = torch.randn(2, 73, 720, 1440, device=dev), torch.randn(2, 73, 720, 1440, device=dev)
def step():                     # bench.py's step: the loss rides in the head's epilogue
    net.zero_grad()
    with loss_obj.fused_with(model, y):
        g = net(x)
    l = loss_obj(g, y, x); l.backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 or e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:70]:
    if not any(k in e.key for k in ("gemm_", "mlp_", "attn_", "proj_ln", "tn_reduce", "ln_")):
        print(f"{e.key[:90]:90s} n={e.count:4d} self_dev={e.self_device_time_total:9.1f} us")

# where the small aten launches come from: python call sites of the ops that launch fill / copy / elementwise kernels
import collections
sites = collections.Counter()
for ev in prof.events():
    if ev.key in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::mul", "aten::add_", "aten::div_", "aten::bernoulli_") and ev.stack:
        fr = [f for f in ev.stack if "swin_v2_weather_amd" in f or "bench.py" in f or "step_ops" in f]
        sites[(ev.key, fr[0] if fr else ev.stack[0])] += 1
print("call sites of the small aten ops (one step):")
for (k, f), n in sites.most_common(30):
    print(f"  {n:3d} x {k:16s} {f}")
