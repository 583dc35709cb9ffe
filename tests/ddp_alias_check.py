#!/usr/bin/env python3
"""GPU box worker for the DDP tests of the HIP model (spawned by tests/test_gpu_parity.py, one process per rank).

  env RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, SWV2_DDP_BACKEND (nccl = RCCL | gloo: CUDA tensors staged through the
  host, so several ranks can share cuda:0), SWV2_DDP_MODE (plain = no DDP | ddp = stock DDP | alias = DDP +
  helpers.enable_ddp_bucket_grads), SWV2_DDP_NFUTURE (0 | 1: MultiStepWrapper rollout, every block's backward node runs
  n_future + 1 times per pass), SWV2_DDP_OPT (sgd | adam | hipadam), SWV2_DDP_OUT (rank 0 saves {losses, params, used} there).

Each rank trains on ITS slice of one fixed global batch (reference train.py:147-148 / DistributedSampler); with the
mean-over-batch loss used here the DDP-averaged gradient of N ranks equals the 1-process gradient on the whole batch, so
the parameters after a few Adam steps must agree."""
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd.networks import swinv2_global as N          # noqa: E402
from swin_v2_weather_amd.networks.helpers import ddp_bucket_plan, ddp_observed_buckets, enable_ddp_bucket_grads, get_model   # noqa: E402

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("SWV2_DDP_BACKEND", "nccl")
mode = os.environ.get("SWV2_DDP_MODE", "alias")
n_future = int(os.environ.get("SWV2_DDP_NFUTURE", "0"))
steps = int(os.environ.get("SWV2_DDP_STEPS", "4"))
GB = 4                                                               # global batch
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if mode != "plain":
    dist.init_process_group(backend, init_method="env://", rank=rank, world_size=world)

params = SimpleNamespace(nettype="swin", img_size=(96, 144), patch_size=4, depth=4, num_heads=4, n_in_channels=5, n_out_channels=5,
                         embed_dim=64, window_ratio=16, drop_path_rate=0.0, full_pos_embed=True, rel_pos=False, mlp_ratio=4.0,
                         activation_ckpt=False, residual=True, n_future=n_future, add_orography=False, add_landmask=False)
torch.manual_seed(7)
m = get_model(params).to(dev).train()
with torch.no_grad():                                                 # LN weights start at 0 (blocks = identity): randomise
    for n_, p in m.named_parameters():
        if n_.endswith("norm1.weight") or n_.endswith("norm2.weight"):
            p.uniform_(0.5, 1.0)
net = m
if mode != "plain":
    cap_mb = float(os.environ.get("SWV2_DDP_CAP_MB", "25"))
    net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=cap_mb)
    if mode == "alias":
        enable_ddp_bucket_grads(net)
# plain SGD: the parameter difference is then linear in the gradient difference (Adam turns a rounding-level difference of a
# near-zero gradient element into a full +-lr step, which makes a parameter comparison meaningless)
optk = os.environ.get("SWV2_DDP_OPT", "sgd")              # sgd | adam (torch) | hipadam (utils/optim.HipAdam)
if optk == "hipadam":
    from swin_v2_weather_amd.utils.optim import HipAdam
    opt = HipAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.95))
elif optk == "adam":
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.95))
else:
    opt = torch.optim.SGD(m.parameters(), lr=0.02)
g = torch.Generator(device="cpu").manual_seed(1)
x = torch.randn(GB, 5, 96, 144, generator=g)
y = torch.randn(GB, 5 * (n_future + 1), 96, 144, generator=g)
lb = GB // world if mode != "plain" else GB
lo = rank * lb if mode != "plain" else 0
x, y = x[lo:lo + lb].to(dev), y[lo:lo + lb].to(dev)
# SWV2_DDP_LOSS=handler: the trainer's loss path -- LossHandler('l2') with its sums in the head epilogue(s) (LossHandler.fused_with; every
# step's head in a rollout); the handler SUMS over the batch (losses.py:196-201), so it is divided by the local batch to keep the DDP
# average equal to the 1-process whole-batch gradient
lh = None
if os.environ.get("SWV2_DDP_LOSS", "mse") == "handler":
    from swin_v2_weather_amd.utils.losses import LossHandler
    lh = LossHandler(SimpleNamespace(n_future=n_future, img_shape_x=96, img_shape_y=144, loss="l2", channel_weights="none", n_out_channels=5,
                                     model_grid_type="equiangular")).to(dev).train()
losses = []
took = 0
for i in range(steps):
    net.zero_grad()
    if lh is not None:
        with lh.fused_with(net, y):
            gen = net(x)
        took += int(lh._fused is not None and (lh._fused.sums is not None or bool(lh._fused.steps)))
        loss = lh(gen, y, x) / lb
    else:
        loss = ((net(x) - y) ** 2).mean()
    loss.backward()
    opt.step()
    if mode != "plain" and world > 1:
        lt = loss.detach().clone()
        dist.all_reduce(lt)
        loss = lt / world
    losses.append(float(loss))
blocks = [b for b in m.modules() if isinstance(b, N.SwinTransformerV2CrBlock)]
used = sum(1 for b in blocks if all(hasattr(p, "_swv2_bv") for p in b.mlp.parameters()))
stuck = sum(1 for b in blocks if b._bv_in_use)
if rank == 0:
    torch.save({"losses": losses, "params": [p.detach().cpu() for p in m.parameters()], "names": [n for n, _ in m.named_parameters()], "used": used, "stuck": stuck, "fused_steps": took,
                "nranks": dist.get_world_size() if mode != "plain" else 1,
                # what the reducer reports after its bucket rebuild vs the plan the cap was chosen with (helpers.ddp_bucket_plan)
                "buckets_observed": ddp_observed_buckets(net) if mode != "plain" else None,
                "buckets_planned": ddp_bucket_plan(m, cap_mb)[0] if mode != "plain" else None}, os.environ["SWV2_DDP_OUT"])
    print(f"rank 0 of {world} ({backend}, {mode}, n_future={n_future}): losses {losses} bucket-view blocks {used}")
if mode != "plain":
    dist.barrier()
    dist.destroy_process_group()
