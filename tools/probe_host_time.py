#!/usr/bin/env python3
"""GPU box: per-step HOST enqueue time vs GPU time of the benchmark step, in sequence after a synchronisation (is the step host- or
GPU-bound? what does the first step after a fence cost?).  usage: python tools/probe_host_time.py [--rel-pos 1] [--local-batch 2]"""
import argparse, os, sys, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--rel-pos", type=int, default=0)
ap.add_argument("--local-batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--profile", type=int, default=0, help="1: cProfile of 10 steady-state steps (host side), top 35 by cumulative time")
x = ap.parse_args()
a = SimpleNamespace(height=720, width=1440, depth=12, heads=8, embed_dim=128, window_ratio=80, drop_path_rate=0.1, rel_pos=x.rel_pos)
dev = torch.device("cuda", 0)
from swin_v2_weather_amd.networks.helpers import get_model
from swin_v2_weather_amd.utils.losses import LossHandler
from swin_v2_weather_amd.utils.optim import HipAdam
torch.manual_seed(333)
model = get_model(bench.model_params(a)).to(dev).train()
lp = SimpleNamespace(n_future=0, img_shape_x=a.height, img_shape_y=a.width, loss="l2", channel_weights="none", n_out_channels=73, model_grid_type="equiangular")
loss_obj = LossHandler(lp).to(dev)
opt = HipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.95))
B = x.local_batch
pool = [(torch.randn(B, 73, a.height, a.width, device=dev), torch.randn(B, 73, a.height, a.width, device=dev)) for _ in range(2)]

def step(i):
    inp, tar = pool[i % 2]
    model.zero_grad()
    with loss_obj.fused_with(model, tar):
        gen = model(inp)
    loss = loss_obj(gen, tar, inp)
    loss.backward()
    opt.step()

for i in range(12):
    step(i)
torch.cuda.synchronize()
for rnd in range(2):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(x.steps + 1)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    host = []
    t00 = time.perf_counter()
    ev[0].record()
    for i in range(x.steps):
        t0 = time.perf_counter()
        step(i)
        ev[i + 1].record()
        host.append(1e3 * (time.perf_counter() - t0))
    t_enq = 1e3 * (time.perf_counter() - t00)
    torch.cuda.synchronize()
    t_all = 1e3 * (time.perf_counter() - t00)
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(x.steps)]
    print(f"round {rnd}: host enqueue of {x.steps} steps {t_enq:.1f} ms, until the GPU is done {t_all:.1f} ms")
    print("  host ms/step:", " ".join(f"{v:.2f}" for v in host))
    print("  gpu  ms/step:", " ".join(f"{v:.2f}" for v in gpu))
if x.profile:
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for i in range(10):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
