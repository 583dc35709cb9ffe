"""CPU, 2 processes over gloo: the trainer's data-parallel step (train.py:265-303 / DDP :186-190 of the reference).

The product model needs an MI355X, so the oracle model/loss are injected through the Trainer's test hooks: what is
exercised here is the trainer's own plumbing -- env:// rendezvous, local batch = batch_size // world_size, DDP wrap,
step order, loss all-reduce -- and the invariant that N-rank averaged gradients equal the 1-rank gradients of the same
global batch (up to the reference's own sum-over-local-batch loss scaling)."""
import os
import socket
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(tmp):
    from swin_v2_weather_amd.utils.YParams import YParams
    p = YParams(os.path.join(ROOT, "swin_v2_weather_amd", "config", "swin.yaml"), "bench_tiny")
    p["img_size"] = [24, 36]
    p["embed_dim"], p["num_heads"], p["window_ratio"], p["depth"] = 16, 2, 4, 2
    p["in_channels"], p["out_channels"] = list(range(4)), list(range(4))
    p["batch_size"] = 4
    p["exp_dir"] = tmp
    p["synthetic_device_pool"] = 0
    p["synthetic_samples_per_year"] = 12
    p["num_data_workers"] = 0
    p["log_to_screen"] = False
    p["rel_pos"] = True
    p["loss"] = "squared geometric l2"
    return p


def _oracle_model(params):
    from oracle import swin_oracle as O
    from swin_v2_weather_amd.networks.helpers import get_model
    torch.manual_seed(7)
    ref = get_model(params)                                  # parameter container (same init on every rank)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    for k in sd:
        if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            sd[k].fill_(0.7)
    cfg = O.SwinCfg.from_params(params)
    cfg.meta_dropout = 0.0                                    # deterministic CPB for the equivalence check
    return O.OracleNet(cfg, sd)


class _OracleLoss(torch.nn.Module):
    def __init__(self, params):
        super().__init__()
        from oracle import swin_oracle as O
        self.O, self.loss = O, params.loss
        self.register_buffer("chw", O.loss_channel_weights(params.loss, params.n_out_channels, params.n_future))

    def forward(self, prd, tar, inp=None):
        return self.O.geometric_l2_loss(prd, tar, self.chw, self.loss)


def _batch(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(n, 4, 24, 36, generator=g), torch.randn(n, 4, 24, 36, generator=g)


def _worker(rank, world, port, tmp, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from swin_v2_weather_amd.train import Trainer
    args = SimpleNamespace(sweep_id=None, config="bench_tiny", run_num="00", enable_amp=False)
    tr = Trainer(_params(tmp), args, model_factory=_oracle_model, loss_factory=_OracleLoss, device="cpu")
    tr.build()
    assert tr.params.local_batch_size == 2 and tr.params.data_num_shards == world
    tr.model.train()
    loss = tr.train_step(_batch(rank, 2))
    dist.all_reduce(loss)
    grads = {n: p.grad.clone() for n, p in tr.model.module.named_parameters()}
    # the epoch loop with the reference's per-step loss synchronisation (log_every_n_steps = 1: all_reduce + .item() every step,
    # train.py:292-294) and with it taken out of the step loop (N = 2: device-side running sum, one all-reduce at the end of the epoch):
    # the same epoch-mean loss from the same start (lr = 0: the parameters stay put)
    for g_ in tr.optimizer.param_groups:
        g_["lr"] = 0.0
    epoch = {}
    for every in (1, 2):
        tr.params["log_every_n_steps"] = every
        if tr.train_sampler is not None:
            tr.train_sampler.set_epoch(0)
        _, _, logs = tr.train_one_epoch()
        epoch[every] = logs
    if rank == 0:
        torch.save({"grads": grads, "loss": float(loss), "epoch": epoch}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_matches_single_rank():
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), tmp, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    # single-process reference on the concatenated global batch
    p = _params(tmp)
    p["n_in_channels"], p["n_out_channels"] = 4, 4
    p["img_shape_x"], p["img_shape_y"] = 24, 36
    model, loss_obj = _oracle_model(p), _OracleLoss(p)
    model.train()
    xs, ts = zip(*[_batch(r, 2) for r in range(2)])
    loss = loss_obj(model(torch.cat(xs)), torch.cat(ts))
    loss.backward()
    assert abs(got["loss"] - float(loss)) <= 1e-5 * abs(float(loss))           # all-reduced sum of local (sum) losses
    e1, e2 = got["epoch"][1], got["epoch"][2]
    assert e1["loss"] > 0 and abs(e1["loss"] - e2["loss"]) <= 1e-6 * abs(e1["loss"])
    assert "last_step_loss" in e1 and "last_step_loss" not in e2               # the per-step value exists only where a step was read back
    for n, prm in model.named_parameters():
        # the loss is a SUM over the local batch and DDP averages over ranks: grad_ddp = grad_global_sum / world
        torch.testing.assert_close(got["grads"][n], prm.grad / 2, rtol=1e-4, atol=1e-6)
