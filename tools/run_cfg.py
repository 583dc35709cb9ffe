#!/usr/bin/env python3
"""Run a few optimisation steps of a named yaml config through the Trainer on one GPU and report ms/step (GPU box).
   python tools/run_cfg.py bench_geo_depth24_e192_invar [local_batch] [steps]"""
import os, sys, time, torch
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd.utils.YParams import YParams
from swin_v2_weather_amd.train import Trainer

cfg = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
p = YParams(os.path.join(ROOT, "swin_v2_weather_amd", "config", "swin.yaml"), cfg)
p["batch_size"], p["max_epochs"] = B, 1
p["synthetic_device_pool"], p["synthetic_steps_per_epoch"] = 2, steps + 3
p["exp_dir"], p["save_checkpoint"], p["log_to_screen"], p["log_to_wandb"] = "/tmp/exp_run_cfg", False, False, False
tr = Trainer(p, SimpleNamespace(sweep_id=None, config=cfg, run_num="00", enable_amp=True))
tr.build()
it = iter(tr.train_data_loader)
losses = []
for i in range(steps + 3):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    losses.append(float(tr.train_step(next(it))))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{cfg}: local batch {B}, {dt * 1e3:.1f} ms/step, {B / dt:.1f} samples/s, losses {losses[0]:.4f} -> {losses[-1]:.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, params {tr.count_parameters() / 1e6:.1f} M")
