#!/usr/bin/env python3
"""Does the model's inference forward capture into a HIP graph (torch.cuda.CUDAGraph), and what does replay buy?  GPU box."""
import os, sys, time, torch
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd.networks import helpers
dev = torch.device("cuda:0")
full = len(sys.argv) > 1 and sys.argv[1] == "full"
H, W = (720, 1440) if full else (144, 288)
p = SimpleNamespace(nettype="swin", img_size=[H, W], patch_size=4, depth=12 if full else 2, num_heads=8, n_in_channels=73, n_out_channels=73,
                    embed_dim=128, window_ratio=80 if full else 16, drop_path_rate=0.0, full_pos_embed=True, rel_pos=False, mlp_ratio=4,
                    activation_ckpt=False, residual=False, n_future=0, add_orography=False, add_landmask=False)
model = helpers.get_model(p).to(dev).eval()
x = torch.randn(1, 73, H, W, device=dev)
with torch.no_grad():
    y0 = model(x).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yg = model(x)
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", bool(torch.equal(yg, y0)), float((yg - y0).abs().max()))

    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f"eager forward {t(lambda: model(x)):.3f} ms, graph replay {t(g.replay):.3f} ms")
