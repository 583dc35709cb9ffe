#!/usr/bin/env python3
"""Headline benchmark: ERA5 samples/sec of the swin_73var depth-12 training step on N MI355X (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W]

N > 1: one process per GPU over RCCL (reference launch contract: train.py:59-68, export_DDP_vars.sh:1-6).  Either the
caller starts the N ranks itself (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, i.e.
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- plain `python bench.py --gpus N` with no WORLD_SIZE --
this process starts them as CHILD processes before it has touched the GPU, relays rank 0's JSON line and exits with the
children's return code.  A rank whose group size differs from --gpus, or that shares its device with another rank under
RCCL, fails with a non-zero exit instead of printing a line for a smaller job.

A "step" is one full optimisation step of the reference's loop (train.py:275-289) on synthetic N(0,1) fields already
resident in HBM: zero_grad -> forward -> geometric l2 loss -> backward (+ RCCL gradient all-reduce under DDP) -> Adam.
Workload at every N: swin_73var, depth 12, embed_dim 128, 8 heads, 73 x 720 x 1440 (the 721-row grid cropped like
data_loader_era5.py:163), local batch 2 per GPU (weak scaling), bf16 MFMA with fp32 accumulation.
Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from types import SimpleNamespace

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument otherwise);
# must be in the environment before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_TOK, L_WIN, PATCH = 180 * 360, 162, 4


def model_params(a):
    return SimpleNamespace(nettype="swin", img_size=[a.height, a.width], patch_size=4, depth=a.depth, num_heads=a.heads,
                           n_in_channels=73, n_out_channels=73, embed_dim=a.embed_dim, window_ratio=a.window_ratio,
                           drop_path_rate=a.drop_path_rate, full_pos_embed=True, rel_pos=bool(a.rel_pos), mlp_ratio=4,
                           activation_ckpt=False, residual=False, n_future=0, add_orography=False, add_landmask=False)


def train_flops_per_sample(a):
    """F_fwd = 2 T P^2 C (Cin + Cout) + depth T (24 C^2 + 4 L C); F_train = 3 F_fwd"""
    T = (a.height // 4) * (a.width // 4)
    Lw = (a.height // a.window_ratio) * (a.width // a.window_ratio)
    C = a.embed_dim
    return 3.0 * (2.0 * T * 16 * C * (73 + 73) + a.depth * T * (24.0 * C * C + 4.0 * Lw * C))


def cpu_baseline(a):
    """The oracle (oracle/swin_oracle.py: the CPU restatement of the reference's model, fp32 torch on the host cores) on a
    BOUNDED sample of the same workload, as SURVEY 8(d) defines it: the REAL depth-12 / 73 x H x W model, one sample, forward +
    backward, 1 warm-up + 2 timed iterations (about 30 s each on this box's host cores, ~31 GB of RAM); optimizer excluded.
    `--cpu-baseline extrapolate` keeps round 2's cheaper estimate (embed + head and embed + 1 block + head, extrapolated
    linearly in the depth) for boxes short of memory or time; it is labelled as such."""
    from oracle import swin_oracle as O
    from swin_v2_weather_amd.networks.helpers import get_model
    torch.manual_seed(0)
    threads = torch.get_num_threads()

    def timed(depth):
        p = model_params(a)
        p.depth = max(depth, 1)
        ref = get_model(p)                                  # parameter container only (CPU); never run on CPU
        sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
        for k in sd:                                        # LN weights are 0 at init: blocks would be the identity
            if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
                sd[k].fill_(1.0)
        cfg = O.SwinCfg.from_params(p)
        cfg.depth = depth
        net = O.OracleNet(cfg, sd)
        x = torch.randn(1, 73, a.height, a.width)
        ts = []
        for it in range(3):                                 # 1 warm-up (allocator, thread pool) + 2 timed
            t0 = time.time()
            y = net(x)
            y.square().mean().backward()
            ts.append(time.time() - t0)
            del y
        del net
        return ts[1:]

    mode = a.cpu_baseline
    if mode == "full":      # the full pass keeps ~31 GB of activations: never drive a box out of memory for a baseline
        try:
            import psutil
            avail = psutil.virtual_memory().available / 2 ** 30
        except Exception:
            avail = 0.0
        if avail < 64.0:
            print(f"bench.py: {avail:.0f} GiB of host memory available, the full-depth CPU baseline wants 64: extrapolating", file=sys.stderr)
            mode = "extrapolate"
    if mode == "extrapolate":
        t0, t1 = min(timed(0)), min(timed(1))
        per_sample = t0 + a.depth * (t1 - t0)
        return {"value": 1.0 / per_sample, "unit": "samples/sec", "cores": threads, "kind": "port", "extrapolated": True,
                "sample": f"oracle fp32 fwd+bwd of 1 sample 73x{a.height}x{a.width}, 1 warm + 2 timed iterations: embed+head "
                          f"{t0:.1f}s, +1 block {t1 - t0:.1f}s, EXTRAPOLATED linearly to depth {a.depth} ({per_sample:.0f}s/sample); "
                          f"optimizer excluded"}
    ts = timed(a.depth)
    per_sample = sum(ts) / len(ts)
    return {"value": 1.0 / per_sample, "unit": "samples/sec", "cores": threads, "kind": "port", "extrapolated": False,
            "sample": f"oracle fp32 forward + backward of 1 sample 73x{a.height}x{a.width} through the full depth-{a.depth} model, "
                      f"1 warm-up + {len(ts)} timed iterations ({', '.join(f'{t:.1f}s' for t in ts)}); optimizer excluded"}


# kernels timed with HIP events inside the timed region (round robin over the 12 blocks of every step):
#   name -> (rocprof kernel-name fragment, bound, algorithmic bytes per token-channel and sample, MFMA flops per T*L*C*B)
ROOFLINE_KERNELS = {
    "attn_bwd": ("attn_bwd_", "hbm", 16.0, 8.0),            # bf16 q, k, v, o, do in; dq, dk, dv out; attn_bwd_stream_kernel or attn_bwd_kernel
    "attn_fwd": ("attn_fwd3_kernel", "hbm", 8.0, 4.0),      # bf16 q, k, v in; o out
    "mlp_bwd": ("mlp_bwd_kernel", "hbm", 28.0, 0.0),        # dx2, x-stats, a2, hpre in; da2, dh, dx1 out
    "mlp_fwd": ("mlp_fwd_kernel", "hbm", 18.0, 0.0),
    # the block's four weight gradients in one launch (gemm_tn_slab.hip): dY + X streams of fc2 (2 + 8), fc1 (8 + 4), proj (2 + 2),
    # qkv (6 + 4) bytes per token-channel; the event bracket also covers the ~12 us reduction of the partial outputs that follows it
    "wgrad_group": ("gemm_tn_slab_c128_kernel", "hbm", 36.0, 0.0),
    # the rest of the attention module: gather + qkv + head split + normalise; proj + LN1 (+ residual, scatter) and
    # its backward; d(qkv) -> dx.  Padded window rows (176 / 162) are in the per-token-channel figures.
    "qkv": ("gemm_rw_kernel<0, 2", "hbm", 10.0, 0.0),
    "proj_ln_fwd": ("proj_ln_fwd_kernel", "hbm", 12.3, 0.0),
    "proj_ln_bwd": ("proj_ln_bwd_kernel", "hbm", 10.5, 0.0),
    "dx": ("gemm_rw_kernel<3, 1", "hbm", 14.0, 0.0),
}
ATTENTION_MODULE = ("qkv", "attn_fwd", "proj_ln_fwd", "proj_ln_bwd", "attn_bwd", "dx")
# SURVEY 8d's bytes for a FUSED module (bf16 x in, y out: 4 B per token-channel forward; x, dy in, dx out (+ y): 8 B backward) -- what the
# module would have to move if nothing were materialised between its kernels.  The per-kernel figures above are this DESIGN's bytes (they
# include the materialised q / k / v, hpre, dh ...): `frac` says how well a kernel moves what it moves, `frac_8d` how far the design is
# from what the module needs to move.  Kernels that are one piece of a module carry the module's name instead.
BYTES_8D = {"mlp_fwd": 4.0, "mlp_bwd": 8.0, "attn_fwd": 8.0, "attn_bwd": 16.0}
MODULE_OF = {"qkv": "attention_module", "proj_ln_fwd": "attention_module", "proj_ln_bwd": "attention_module", "dx": "attention_module",
             "wgrad_group": "weight gradients (SURVEY 8d gives no byte figure: dW = dY^T X streams both operands once, the design's 36 B)"}


def roofline_entry(name, ktimes, a, pmc, B):
    frag, bound, bpe, fpe = ROOFLINE_KERNELS[name]
    T = (a.height // 4) * (a.width // 4)
    Lw = (a.height // a.window_ratio) * (a.width // a.window_ratio)
    n_l, k_ms = ktimes.get(name, (0, 0.0))
    alg = bpe * T * a.embed_dim * B
    achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # counter traffic: only from a profile of THESE kernel sources (hash recorded by profiles/summarize.py), at this local batch,
    # and only when exactly one profiled kernel carries the name fragment -- otherwise null with the reason
    traffic, src = None, None
    if pmc is not None:
        from swin_v2_weather_amd import _lib as L_
        hits = [k for k, v in pmc[1].items() if frag in k and isinstance(v, dict) and "hbm_bytes_per_launch" in v]
        if pmc[1].get("_source_hash") != L_.source_hash():
            src = f"{pmc[0]}: taken from other kernel sources than the library being timed -- not reported"
        elif B != pmc[1].get("_local_batch", 2):
            src = f"{pmc[0]}: profiled at local batch {pmc[1].get('_local_batch', 2)} -- not reported"
        elif len(hits) != 1:
            src = f"{pmc[0]}: {len(hits)} profiled kernels match '{frag}' -- not reported"
        else:
            traffic, src = pmc[1][hits[0]]["hbm_bytes_per_launch"], f"{pmc[0]}: {hits[0]}"
    e = {"kernel": name, "bound": bound, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
         "traffic": traffic, "traffic_source": src, "launches_timed": n_l, "avg_ms": k_ms, "algorithmic_bytes_per_launch": alg,
         "design_bytes": alg}
    if name in BYTES_8D:
        a8 = BYTES_8D[name] * T * a.embed_dim * B
        e["algorithmic_bytes_8d"] = a8
        e["frac_8d"] = a8 / (k_ms * 1e-3) / 1e9 / 8000.0 if k_ms > 0 else 0.0
    else:
        e["algorithmic_bytes_8d"], e["frac_8d"], e["part_of"] = None, None, MODULE_OF.get(name)
    if fpe:
        fl = fpe * T * Lw * a.embed_dim * B
        e["flops_per_launch"] = fl
        e["mfma_frac"] = fl / (k_ms * 1e-3) / 2.5e15 if k_ms > 0 else 0.0
    return e


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks as child processes of THIS process, which has not
    made a single HIP call (importing torch does not initialise the GPU; no exec of a GPU-initialised process anywhere), relay
    rank 0's one JSON line and return the children's exit code."""
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
    for l in lines[-1:]:
        print(l, flush=True)
    if r.returncode == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    return r.returncode


def attention_module(ktimes, a, B):
    """qkv + attention core + proj / LN1, forward + backward data path (fwd T (8 C^2 + 4 L C), x3 with the backward;
    the weight gradients of qkv / proj are inside the grouped launch and not separable): MFMA flops over the event-timed launch
    durations, and the counter-based pipe utilisation of the same kernels from the committed profile (null when the profile
    was taken from other sources)."""
    T = (a.height // 4) * (a.width // 4)
    Lw = (a.height // a.window_ratio) * (a.width // a.window_ratio)
    C = a.embed_dim
    t_ms = sum(ktimes.get(k, (0, 0.0))[1] for k in ATTENTION_MODULE)
    if t_ms <= 0 or any(ktimes.get(k, (0, 0.0))[0] == 0 for k in ATTENTION_MODULE):
        return None
    # data path only: forward 8 C^2 + 4 L C per token, backward the same again for dX (dW excluded) -> x2
    flops = 2.0 * T * (8.0 * C * C + 4.0 * Lw * C) * B
    # SURVEY 8d, fused module: forward 4 T C, backward 8 T C bytes (bf16 x -> y; x, dy -> dx)
    a8 = 12.0 * T * C * B
    out = {"kernels": list(ATTENTION_MODULE), "ms_per_block": t_ms, "data_path_flops_per_block": flops,
           "mfma_frac_by_flops": flops / (t_ms * 1e-3) / 2.5e15, "mfma_pipe_busy_frac": None, "counter_source": None,
           "algorithmic_bytes_8d": a8, "frac_8d": a8 / (t_ms * 1e-3) / 1e9 / 8000.0,
           "design_bytes": sum(ROOFLINE_KERNELS[k][2] for k in ATTENTION_MODULE) * T * C * B}
    try:
        import glob
        from swin_v2_weather_amd import _lib as L_
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))[-1]
        d = json.load(open(f))
        frags = [ROOFLINE_KERNELS[k][0] for k in ATTENTION_MODULE]
        hits = [[n for n in d["kernels"] if fr in n] for fr in frags]
        if d.get("_source_hash") == L_.source_hash() and all(len(h) == 1 for h in hits):
            busy = sum(d["kernels"][h[0]]["SQ_VALU_MFMA_BUSY_CYCLES"] for h in hits)
            simd = sum(d["kernels"][h[0]]["SQ_BUSY_CYCLES"] * 32.0 for h in hits)
            out["mfma_pipe_busy_frac"], out["counter_source"] = busy / simd, os.path.relpath(f, ROOT)
        else:
            out["counter_source"] = f"{os.path.relpath(f, ROOT)}: other kernel sources or ambiguous kernel names -- not reported"
    except Exception:
        pass
    return out


def secondary_legs(a, dev, rank):
    """After the timed region (never part of `value`): the other BASELINE.json configurations and the variants VERDICT r4 asked to see
    in the driver-run line instead of builder-written files -- each 3 settle + 8 timed steps of the same full optimisation step, same
    process, one GPU: the CPB-bias model (rel_pos=True: the attention the north star describes), configs[2]'s per-GPU load (local batch
    8), configs[3] (depth 24 / embed 192 / 77 input channels with zenith + invariants / channel-weighted loss, through the Trainer) and
    configs[4] (2-step autoregressive rollout, through the Trainer).  Returns a list of {"workload", "local_batch", "value",
    "ms_per_step"} (+ "error" when a leg failed: a broken leg must not cost the headline line)."""
    import gc
    import tempfile
    from swin_v2_weather_amd.networks.helpers import get_model
    from swin_v2_weather_amd.utils.losses import LossHandler
    from swin_v2_weather_amd.utils.optim import HipAdam
    settle, steps = 3, 8
    out = []

    def timed(step_fn):
        for i in range(settle):
            step_fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(settle + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def bench_loop_leg(label, rel_pos, B):
        torch.manual_seed(333)
        p = model_params(a)
        p.rel_pos = bool(rel_pos)
        model = get_model(p).to(dev)
        model.train()
        lp = SimpleNamespace(n_future=0, img_shape_x=a.height, img_shape_y=a.width, loss="l2", channel_weights="none",
                             n_out_channels=73, model_grid_type="equiangular")
        loss_obj = LossHandler(lp).to(dev)
        opt = HipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.95))
        g = torch.Generator(device=dev).manual_seed(333 + rank)
        pool = [(torch.randn(B, 73, a.height, a.width, device=dev, generator=g),
                 torch.randn(B, 73, a.height, a.width, device=dev, generator=g)) for _ in range(2)]

        def step(i):
            inp, tar = pool[i % 2]
            model.zero_grad()
            with loss_obj.fused_with(model, tar):
                gen = model(inp)
            loss = loss_obj(gen, tar, inp)
            loss.backward()
            opt.step()
        dt = timed(step)
        return {"workload": label, "local_batch": B, "value": B / dt, "ms_per_step": 1e3 * dt, "loop": "bench.py"}

    def trainer_leg(label, cfg_name, B):
        from swin_v2_weather_amd.train import Trainer
        from swin_v2_weather_amd.utils.YParams import YParams
        p = YParams(os.path.join(ROOT, "swin_v2_weather_amd", "config", "swin.yaml"), cfg_name)
        p["batch_size"], p["max_epochs"] = B, 1
        p["synthetic_device_pool"], p["synthetic_steps_per_epoch"] = 2, settle + steps
        p["exp_dir"], p["save_checkpoint"], p["log_to_screen"], p["log_to_wandb"] = tempfile.mkdtemp(prefix="swv2_bench_"), False, False, False
        tr = Trainer(p, SimpleNamespace(sweep_id=None, config=cfg_name, run_num="00", enable_amp=True))
        tr.build()
        tr.model.train()
        it = iter(tr.train_data_loader)
        dt = timed(lambda i: tr.train_step(next(it)))
        return {"workload": label, "local_batch": B, "value": B / dt, "ms_per_step": 1e3 * dt, "loop": "Trainer.train_step", "yaml": cfg_name}

    base = f"swin_73var depth{a.depth} embed{a.embed_dim} heads{a.heads} 73x{a.height}x{a.width}"
    legs = [
        (lambda: bench_loop_leg(base + " rel_pos=True (log-spaced CPB bias, meta-MLP dropout on)", 1, a.local_batch)),
        (lambda: bench_loop_leg(base + " rel_pos=False at configs[2]'s per-GPU load", 0, 8)),
        (lambda: trainer_leg("configs[3]: swin_73var_geo depth24 embed192 77ch (zenith + orography + landmask) chweight loss", "bench_geo_depth24_e192_invar", 2)),
        (lambda: trainer_leg("configs[4]: 2-step autoregressive finetune, depth12 embed128 77ch, multi-step loss", "bench_depth12_e128_2step", 2)),
    ]
    for leg in legs:
        try:
            out.append(leg())
        except Exception as e:                      # noqa: BLE001
            out.append({"workload": "leg failed", "error": f"{type(e).__name__}: {e}"[:300]})
        gc.collect()
        torch.cuda.empty_cache()
    return out


def ddp_extra_legs(a, net, step, fence, dev, rank, world, B, make_pool, loss_obj, opt):
    """After the timed region of an N > 1 run (never part of `value`):
    (1) `exposed_comm`: K more steps inside `net.no_sync()` -- same kernels, no gradient all-reduce; ms_per_step(DDP) - ms_per_step(no_sync) is
        the communication that backward did not hide (the reference overlaps bucketed all-reduces with backward: train.py:186-190);
    (2) BASELINE configs[2]: the same model under the same DDP wrapper at local batch 8 (global batch 64 at 8 GPUs), timed like the headline
        (barrier + synchronize on both sides, max over ranks).  The headline stays local batch 2 at every N, so that the N = 1 point of a
        scaling run is the BENCH line and per-GPU work is fixed (weak scaling); this leg is the same measurement at configs[2]'s load."""
    K = max(4, min(a.steps, 10))

    def timed(fn, n):
        fence()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        fence()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        return max(float(x) for x in ts) / n, min(float(x) for x in ts) / n
    for i in range(2):
        step(i)
    t_ddp, _ = timed(step, K)
    with net.no_sync():
        for i in range(2):
            step(i)
        t_ns, _ = timed(step, K)
    out = {"exposed_comm": {"ms_per_step_ddp": 1e3 * t_ddp, "ms_per_step_no_sync": 1e3 * t_ns, "exposed_ms": 1e3 * max(0.0, t_ddp - t_ns),
                            "steps": K, "method": "same steps under DistributedDataParallel.no_sync(), max over ranks"}}
    B2 = 8
    if B != B2 and not a.no_secondary:
        try:
            pool2 = make_pool(B2)

            def step2(i):
                inp, tar = pool2[i % len(pool2)]
                net.zero_grad()
                with loss_obj.fused_with(net, tar):
                    gen = net(inp)
                loss = loss_obj(gen, tar, inp)
                loss.backward()
                opt.step()
            for i in range(4):
                step2(i)
            t2, t2min = timed(step2, K)
            with net.no_sync():
                for i in range(2):
                    step2(i)
                t2ns, _ = timed(step2, K)
            out["secondary"] = [{"workload": f"BASELINE configs[2]: swin_73var depth{a.depth} embed{a.embed_dim}, DDP over {world} ranks ({a.backend}), "
                                             f"local batch {B2} (global {B2 * world})", "local_batch": B2, "global_batch": B2 * world,
                                 "value": world * B2 / t2, "unit": "samples/sec", "ms_per_step": 1e3 * t2,
                                 "rank_ms_per_step": {"min": 1e3 * t2min, "max": 1e3 * t2}, "steps": K,
                                 "exposed_comm_ms": 1e3 * max(0.0, t2 - t2ns), "ms_per_step_no_sync": 1e3 * t2ns}]
        except Exception as e:                      # noqa: BLE001
            out["secondary"] = [{"workload": "configs[2] leg failed", "error": f"{type(e).__name__}: {e}"[:300]}]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--local-batch", type=int, default=2)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--embed-dim", type=int, default=128)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1440)
    ap.add_argument("--window-ratio", type=int, default=80)
    ap.add_argument("--rel-pos", type=int, default=0, help="0 = yaml default of swin_73var (rel_pos: false)")
    ap.add_argument("--drop-path-rate", type=float, default=0.1)
    ap.add_argument("--pool", type=int, default=2, help="device-resident synthetic batches that are cycled")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DDP / init RCCL even with one rank (self-test)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the measured path); gloo: self-test of the N > 1 code path with all ranks on one GPU "
                         "(tests/test_gpu_parity.py), never a benchmark")
    ap.add_argument("--time-every", type=int, default=6, help="bracket every N-th launch of the roofline kernel inside the timed region with HIP events")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: no HIP event pairs inside the timed region (the roofline object is then empty)")
    ap.add_argument("--settle", type=int, default=8, help="untimed set-up steps before the W warm-up steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (rel_pos=True, local batch 8, configs[3], configs[4]) that follow the timed region at N = 1")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "extrapolate"],
                    help="full: the real depth-D model, 1 warm-up + 2 timed iterations (default); extrapolate: depth 0 and 1 only")
    ap.add_argument("--roofline-kernel", default="attn_bwd")
    ap.add_argument("--data", default="resident", choices=["resident", "host"],
                    help="host: after the timed (HBM-resident) region, time the same step fed from HOST-resident fields through the "
                         "input pipeline (pinned staging, async H2D, assembly kernels) and report it as `host_pipeline`")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:        # no launcher: become the launcher (before any GPU call)
        raise SystemExit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to print a line for a different job size")
    n_dev = torch.cuda.device_count()                        # (counting devices does not initialise the GPU)
    if a.backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {n_dev} (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    dev_index = local_rank % n_dev                           # gloo self-test only: several ranks share the one GPU of the box
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_ddp = world > 1 or a.force_ddp
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner to (C-buffered) stdout when the communicator is created; the bench contract is ONE
        # JSON line on stdout, so fd 1 points at stderr until the communicator exists
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if a.backend == "nccl":
                dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)      # nccl = RCCL on ROCm
            else:
                dist.init_process_group(backend="gloo", init_method="env://")
            dist.barrier()
            torch.cuda.synchronize()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    from swin_v2_weather_amd import ops
    from swin_v2_weather_amd.networks.helpers import get_model
    from swin_v2_weather_amd.utils.losses import LossHandler

    torch.manual_seed(333)                                   # same initial weights on every rank
    p = model_params(a)
    model = get_model(p).to(dev)
    model.train()
    lp = SimpleNamespace(n_future=0, img_shape_x=a.height, img_shape_y=a.width, loss="l2", channel_weights="none",
                         n_out_channels=73, model_grid_type="equiangular")
    loss_obj = LossHandler(lp).to(dev)
    from swin_v2_weather_amd.utils.optim import HipAdam
    opt = HipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.95))        # train.py:176, one swv2_adam_multi launch per step
    net = model
    from swin_v2_weather_amd.networks.helpers import DDP_BUCKET_CAP_MB, ddp_bucket_plan
    if use_ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_index], output_device=dev_index,
                                                        broadcast_buffers=False, gradient_as_bucket_view=True,
                                                        bucket_cap_mb=DDP_BUCKET_CAP_MB)    # >= 4 block buckets during backward, pos_embed closes its own (helpers.py)
        if os.environ.get("SWV2_DDP_BUCKET_GRADS", "1") != "0":
            from swin_v2_weather_amd.networks.helpers import enable_ddp_bucket_grads
            enable_ddp_bucket_grads(net)
    g = torch.Generator(device=dev).manual_seed(333 + rank)
    B = a.local_batch
    def make_pool(b):
        return [(torch.randn(b, 73, a.height, a.width, device=dev, generator=g),
                 torch.randn(b, 73, a.height, a.width, device=dev, generator=g)) for _ in range(a.pool)]
    pool = make_pool(B)

    def step(i):
        inp, tar = pool[i % len(pool)]
        net.zero_grad()
        with loss_obj.fused_with(net, tar):      # as train.py's step: the loss sums ride in the head GEMM's epilogue
            gen = net(inp)
        loss = loss_obj(gen, tar, inp)
        loss.backward()
        opt.step()
        return loss

    def fence():
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.settle):          # untimed set-up: the caching allocator grows to its steady-state footprint and the
        step(i)                        # clocks settle during the first ~10 steps (one of them stalls for ~80 ms)
    # Every HIP event the timed region uses is created (and recorded once: that is what materialises its handle) BEFORE the warm-up
    # steps, so that nothing but the barrier + synchronize the contract asks for sits between the last warm-up step and the first timed
    # one: a GPU left idle for a few milliseconds re-ramps its clocks, and the first timed step (10.4 - 14.7 ms against 9.1, round 5
    # measurements) is already the one that cannot overlap its host-side launch work with a previous step's kernels.
    #   * one event per STEP boundary (K + 1 records on the launch stream): the GPU-side duration of every timed step, so that a
    #     reader can tell box noise from signal (the timed region is 0.2 s);
    #   * inside the timed region only the roofline kernel is bracketed with HIP events, and only on every 6th eligible block call
    #     (2 of its 12 launches per step): an event record between two kernels is a packet the next kernel waits behind -- one pair per
    #     block call (24 per step, what rounds 2 - 3 did to time nine kernels at once) measured 3.3 % of the step (208.5 vs 215.5
    #     samples/s).  The other kernels of `roofline_others` are timed in extra steps AFTER the timed region.
    ops.prewarm_events(64)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    for e_ in step_ev:
        e_.record()
    for i in range(a.warmup):
        step(i)
    fence()
    ops.start_kernel_timing([] if a.no_kernel_timing else [a.roofline_kernel], every=a.time_every)
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(a.steps):
        loss = step(a.warmup + i)
        step_ev[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    ktimes = ops.stop_kernel_timing()
    step_seq = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(a.steps)]
    step_ms = sorted(step_seq)
    others = [k for k in ROOFLINE_KERNELS if k != a.roofline_kernel]
    if others and not a.no_kernel_timing:          # not part of `value`: three more steps with every block call bracketed
        ops.start_kernel_timing(others)
        for i in range(3):
            step(i)
        fence()
        ktimes.update(ops.stop_kernel_timing())
    # after the timed region (not part of `value`): the same kernel without the co-running weight-gradient stream
    alone_ms = None
    if world == 1 and os.environ.get("SWV2_WGRAD_SIDE_STREAM", "0") != "0":
        for m_ in model.modules():
            for r_ in getattr(m_, "_runners", {}).values():
                r_.desc.wgrad_side_stream = 0
        ops.start_kernel_timing([a.roofline_kernel])
        for i in range(3):
            step(i)
        fence()
        alone_ms = ops.stop_kernel_timing().get(a.roofline_kernel, (0, None))[1]
    host_leg = None
    if a.data == "host":
        # PCIe-inclusive rate (never `value`): host-resident fp32 fields [N, 73, 721, 1440] -> staging ring -> H2D -> assembly
        from swin_v2_weather_amd.utils.host_pipeline import Era5HostPipeline, SyntheticYearSource

        class P(dict):
            __getattr__ = dict.__getitem__
        host_leg = {}
        for label, pinned in (("staged", False), ("zero_copy_pinned_source", True)):
            src = SyntheticYearSource(n_years=1, n_samples=6, seed=333 + rank, pinned=pinned)
            hp_params = P(local_batch_size=B, dt=1, n_future=0, img_size=(a.height, a.width), in_channels=list(range(73)),
                          out_channels=list(range(73)), add_zenith=False, seed=333, data_num_shards=1, data_shard_id=0,
                          num_data_workers=16)
            pipe = Era5HostPipeline(hp_params, src, dev, train=True, steps_per_epoch=a.steps + 2)
            fence()
            n_done, t1 = 0, None
            for bi, (inp, tar, _) in enumerate(pipe):
                if bi == 2:                                  # two untimed steps fill the ring
                    fence()
                    t1 = time.perf_counter()
                net.zero_grad()
                with loss_obj.fused_with(net, tar):
                    gen_h = net(inp)
                loss_h = loss_obj(gen_h, tar, inp)
                loss_h.backward()
                opt.step()
                n_done += int(bi >= 2)
            fence()
            dth = time.perf_counter() - t1
            host_leg[label] = {"samples_per_s": world * B * n_done / dth, "ms_per_step": 1e3 * dth / n_done,
                               "h2d_gb_per_s_per_gpu": B * n_done * 2 * 73 * 721 * 1440 * 4 / dth / 1e9}
            del pipe, src
    final_loss = float(loss.detach())
    ddp_observed = None
    ddp_legs = None
    if use_ddp:
        from swin_v2_weather_amd.networks.helpers import ddp_observed_buckets
        ddp_observed = ddp_observed_buckets(net)
        ddp_legs = ddp_extra_legs(a, net, step, fence, dev, rank, world, B, make_pool, loss_obj, opt)
    secondary = None
    if world == 1 and not use_ddp and not a.no_secondary:
        # the main model / pool are no longer needed: free them before the secondary configurations allocate theirs
        del loss, pool, opt, net, model, loss_obj
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        secondary = secondary_legs(a, dev, rank)
    rank_ms = [1e3 * dt / a.steps]
    if use_ddp:
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
        # every rank's own step time and device: value uses the MAX; two RCCL ranks on one device are an error
        mine = torch.tensor([dt, float(dev_index)], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_ms = [1e3 * float(t_[0]) / a.steps for t_ in allr]
        devs = [int(t_[1]) for t_ in allr]
        if a.backend == "nccl" and len(set(devs)) != world:
            raise SystemExit(f"bench.py: ranks share devices {devs}")
        dt = max(float(t_[0]) for t_ in allr)
    if rank == 0:
        value = world * B * a.steps / dt
        flops = train_flops_per_sample(a)
        pmc = None
        try:        # HBM bytes per launch from the committed rocprofv3 --pmc passes (2 x FETCH_SIZE + WRITE_SIZE, see profiles/);
            import glob   # measured at local batch 2 for the kernels of that round -- reported with its source, null otherwise
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))[-1]
            pmc = (os.path.relpath(f, ROOT), json.load(open(f)))
        except Exception:
            pmc = None
        main_rf = roofline_entry(a.roofline_kernel, ktimes, a, pmc, B)
        main_rf["co_running"] = None
        # (the kernel the north star names -- not the largest by time: profiles/rNN_kernel_stats.md ranks the grouped weight gradient
        # and the fused MLP backward above it; their roofline objects are in `roofline_others`)
        main_rf["why_this_kernel"] = "window-attention backward: the kernel BASELINE.json's north_star names; third by time per step"
        main_rf["timed"] = f"HIP event pairs on the launch stream around every {a.time_every}th launch inside the timed region"
        if alone_ms:
            main_rf["avg_ms_alone"] = alone_ms
            main_rf["frac_alone"] = main_rf["algorithmic_bytes_per_launch"] / (alone_ms * 1e-3) / 1e9 / 8000.0
        out = {
            "metric": "ERA5 samples/sec (73x721x1440) swin_73var depth12", "value": value, "unit": "samples/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"swin_73var depth{a.depth} embed{a.embed_dim} heads{a.heads} 73x{a.height}x{a.width} "
                                   f"window{a.height // a.window_ratio}x{a.width // a.window_ratio} rel_pos={bool(a.rel_pos)} "
                                   f"full train step (fwd+loss+bwd+Adam)",
                       "local_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "final_loss": final_loss},
            "model_tflops_per_gpu": value * flops / world / 1e12,
            "mfma_frac_end_to_end": value * flops / world / 2.5e15,
            "host_pipeline": host_leg,
            "backend": a.backend if use_ddp else None,
            "rccl_nranks": dist.get_world_size() if use_ddp else None,
            # what the reducer REPORTS after its bucket rebuild (launch order = gradient arrival), next to the plan the cap was chosen with
            "ddp_buckets_mb": ddp_observed if use_ddp else None,
            "ddp_buckets_mb_planned": ddp_bucket_plan(model)[0] if use_ddp else None,
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
            # GPU-side duration of every timed step of rank 0 (HIP events at the step boundaries): spread of the K steps behind `value`
            "step_ms": {"min": step_ms[0], "p50": step_ms[len(step_ms) // 2], "p90": step_ms[min(len(step_ms) - 1, int(0.9 * len(step_ms)))],
                        "max": step_ms[-1], "n": len(step_ms), "sequence": [round(v, 3) for v in step_seq]},
            "secondary": secondary if secondary is not None else (ddp_legs or {}).get("secondary"),
            # N > 1 only: the same step with the gradient all-reduce switched off (DistributedDataParallel.no_sync) -- the difference is the
            # communication the backward did NOT hide (+ the reducer's own bookkeeping)
            "exposed_comm": (ddp_legs or {}).get("exposed_comm"),
            "weak_scaling_local_batch": B,      # fixed per GPU at every N (BASELINE cfg 2's batch; cfg 3's 8 per GPU: --local-batch 8)
            "roofline": main_rf,
            "roofline_others": [dict(roofline_entry(k, ktimes, a, pmc, B), timed="every launch of 3 extra steps AFTER the timed region")
                                for k in ROOFLINE_KERNELS if k != a.roofline_kernel],
            "attention_module": attention_module(ktimes, a, B),
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out), flush=True)
    if use_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
