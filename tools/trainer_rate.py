#!/usr/bin/env python3
"""GPU box: samples/s of the drop-in Trainer's own epoch loop (swin_v2_weather_amd/train.py::Trainer.train_one_epoch -- what
`python -m swin_v2_weather_amd.train --config <cfg>` runs) next to bench.py's loop, with the reference's per-step loss synchronisation
(log_every_n_steps = 1: all_reduce + .item() every step, train.py:292-294) and with it taken out of the step loop (N = 10).
   python tools/trainer_rate.py [config] [steps]"""
import os, sys, time, tempfile, torch
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd.utils.YParams import YParams
from swin_v2_weather_amd.train import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "bench_depth12_e128"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for every in (1, 10):
    p = YParams(os.path.join(ROOT, "swin_v2_weather_amd", "config", "swin.yaml"), cfg)
    p["max_epochs"], p["synthetic_device_pool"], p["synthetic_steps_per_epoch"] = 1, 2, steps
    p["exp_dir"], p["save_checkpoint"], p["log_to_screen"], p["log_to_wandb"] = tempfile.mkdtemp(prefix="swv2_rate_"), False, False, False
    p["log_every_n_steps"] = every
    tr = Trainer(p, SimpleNamespace(sweep_id=None, config=cfg, run_num="00", enable_amp=True))
    tr.build()
    tr.params["synthetic_steps_per_epoch"] = steps
    tr.train_one_epoch()                         # warm-up epoch (allocator, clocks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr_time, _, logs = tr.train_one_epoch()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    B = int(tr.params.local_batch_size)
    print(f"{cfg}: Trainer.train_one_epoch, {steps} steps, local batch {B}, log_every_n_steps={every}: {logs['samples_per_sec']:.1f} samples/s "
          f"(loop clock), {B * steps / wall:.1f} samples/s (wall incl. the final synchronize), mean loss {logs['loss']:.4f}")
    del tr
    torch.cuda.empty_cache()
