#!/usr/bin/env python3
"""Why is the first timed step after bench.py's fence slow (11.2 ms against 9.15)?  GPU box.  The bench's step in a loop with a fence
(torch.cuda.synchronize) in front of EVERY step, GPU duration per step from events, host time per step from perf_counter, in variants:
  steady      no fences (reference)
  fenced      fence before every step
  zero_late   fence before every step, zero_grad at the END of the previous step instead of at the start of this one
  spin        fence, then a 3 ms host spin BEFORE launching (GPU idle longer: clocks?)"""
import os, sys, time
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from swin_v2_weather_amd.networks.helpers import get_model  # noqa: E402
from swin_v2_weather_amd.utils.losses import LossHandler  # noqa: E402
from swin_v2_weather_amd.utils.optim import HipAdam  # noqa: E402

a = SimpleNamespace(height=720, width=1440, depth=12, heads=8, embed_dim=128, window_ratio=80, drop_path_rate=0.1, rel_pos=0)
dev = torch.device("cuda:0")
torch.manual_seed(333)
model = get_model(bench.model_params(a)).to(dev).train()
loss_obj = LossHandler(SimpleNamespace(n_future=0, img_shape_x=720, img_shape_y=1440, loss="l2", channel_weights="none", n_out_channels=73,
                                       model_grid_type="equiangular")).to(dev)
opt = HipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.95))
pool = [(torch.randn(2, 73, 720, 1440, device=dev), torch.randn(2, 73, 720, 1440, device=dev)) for _ in range(2)]


def step(i, zero_first=True, zero_last=False):
    inp, tar = pool[i % 2]
    if zero_first:
        model.zero_grad()
    with loss_obj.fused_with(model, tar):
        gen = model(inp)
    loss = loss_obj(gen, tar, inp)
    loss.backward()
    opt.step()
    if zero_last:
        model.zero_grad()


for i in range(12):
    step(i)
torch.cuda.synchronize()


def run(name, n=12, fence=False, zero_late=False, spin=0.0):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    host = []
    if zero_late:
        model.zero_grad()
    for i in range(n):
        if fence:
            torch.cuda.synchronize()
        if spin:
            t_ = time.perf_counter()
            while time.perf_counter() - t_ < spin:
                pass
        t0 = time.perf_counter()
        ev[2 * i].record()
        step(i, zero_first=not zero_late, zero_last=zero_late)
        ev[2 * i + 1].record()
        host.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    g = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n))
    print(f"{name:10s} GPU ms per step: min {g[0]:.2f} p50 {g[n // 2]:.2f} max {g[-1]:.2f}   host enqueue ms: p50 {sorted(host)[n // 2]:.2f}")


run("steady")
run("fenced", fence=True)
run("zero_late", fence=True, zero_late=True)
run("spin", fence=True, spin=3e-3)
run("steady")
