import os, sys, time, cProfile, pstats
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from types import SimpleNamespace
a = SimpleNamespace(height=720, width=1440, depth=12, heads=8, embed_dim=128, window_ratio=80, drop_path_rate=0.1, rel_pos=0)
from swin_v2_weather_amd.networks.helpers import get_model
from swin_v2_weather_amd.utils.losses import LossHandler
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = get_model(bench.model_params(a)).to(dev).train()
lp = SimpleNamespace(n_future=0, img_shape_x=720, img_shape_y=1440, loss="l2", channel_weights="none", n_out_channels=73, model_grid_type="equiangular")
loss_obj = LossHandler(lp).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.95), fused=True)
inp = torch.randn(2, 73, 720, 1440, device=dev); tar = torch.randn(2, 73, 720, 1440, device=dev)
def step():
    model.zero_grad()
    gen = model(inp)
    loss = loss_obj(gen, tar, inp)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
def fwd_only():
    t0 = time.perf_counter(); gen = model(inp); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fwd host {1e3*(t1-t0):.1f} total {1e3*(t2-t0):.1f}")
    return gen
gen = fwd_only(); gen = fwd_only()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
print("---- per-step GPU times over 24 steps (events)")
evs = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
host = []
evs[0].record()
for i in range(24):
    t0 = time.perf_counter(); step(); host.append(1e3 * (time.perf_counter() - t0)); evs[i + 1].record()
torch.cuda.synchronize()
print("gpu ms :", " ".join(f"{evs[i].elapsed_time(evs[i+1]):.1f}" for i in range(24)))
print("host ms:", " ".join(f"{h:.1f}" for h in host))
import gc; print("gc counts", gc.get_count(), "mem", torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9)
