#!/usr/bin/env python3
"""GPU box, one rank over RCCL: the DDP bucket-view gradient path (helpers.enable_ddp_bucket_grads) gives the same losses
and parameters as plain DDP and as no DDP, and the per-parameter copy kernels are gone (counted with the profiler)."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
from swin_v2_weather_amd.networks import swinv2_global as N
from swin_v2_weather_amd.networks.helpers import enable_ddp_bucket_grads
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://")

def run(mode, steps=6):
    torch.manual_seed(7)
    m = N.SwinTransformerV2Cr(img_size=(96, 144), patch_size=4, depths=(4,), num_heads=(4,), in_chans=5, out_chans=5, embed_dim=64,
                              img_window_ratio=16, full_pos_embed=True, rel_pos=False, residual=True, drop_path_rate=0.0).to(dev).train()
    net = m
    if mode != "plain":
        net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True)
        if mode == "alias":
            enable_ddp_bucket_grads(net)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.95), fused=True)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(2, 5, 96, 144, device=dev, generator=g); y = torch.randn(2, 5, 96, 144, device=dev, generator=g)
    losses = []
    for i in range(steps):
        net.zero_grad()
        loss = ((net(x) - y) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    used = sum(1 for b in m.modules() if isinstance(b, N.SwinTransformerV2CrBlock) and all(hasattr(p, "_swv2_bv") for p in b.mlp.parameters()))
    return losses, [p.detach().clone() for p in m.parameters()], used

ref_l, ref_p, _ = run("plain")
ddp_l, ddp_p, _ = run("ddp")
al_l, al_p, used = run("alias")
print("losses plain", ref_l); print("losses ddp  ", ddp_l); print("losses alias", al_l, "blocks with cached bucket views:", used)
worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(al_p, ddp_p))
worst2 = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(al_p, ref_p))
print("max rel param diff alias vs ddp %.3e, alias vs plain %.3e" % (worst, worst2))
assert used == 4 and worst < 2e-3 and worst2 < 2e-3 and all(abs(a - b) < 2e-4 * abs(b) for a, b in zip(al_l, ddp_l))
print("ddp alias check ok")
dist.destroy_process_group()
