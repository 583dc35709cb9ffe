"""Registry checkpoints and autoregressive inference on the HIP model.

The reference publishes its models as registry folders (README.md:32-44)

    swin_73var_depth12_chweight_invar/{hyperparams.yaml, weights.tar, global_means.npy, global_stds.npy, metadata.json}

where `hyperparams.yaml` is the flat dump of the training `params` (train.py:156-163) and `weights.tar` is the training
checkpoint (train.py:374-378): `{'model_state': state_dict, ...}` saved from the DDP-wrapped wrapper, i.e. keys
`module.model.<swin key>`.  `load_registry_model` builds the model of that yaml behind the same `get_model` surface and
loads the weights with any of the prefixes the reference produces (`module.model.`, `model.`, none);
`rollout` is the inference loop of MultiStepWrapper (helpers.py:26-41) for an arbitrary number of steps without autograd:
prediction fed back, next cos-zenith channel and the invariant channels re-appended.

    python -m swin_v2_weather_amd.inference --registry DIR --steps 8 [--init x0.npy] [--out forecast.npy]
"""
from __future__ import annotations

import argparse
import os
from types import SimpleNamespace

import numpy as np
import torch

from .networks.helpers import get_model
from .utils.YParams import load_yaml


class _Params(dict):
    """flat hyperparams.yaml -> the attribute + item access train.py's `params` offers"""
    __getattr__ = dict.__getitem__

    def __contains__(self, k):
        return dict.__contains__(self, k)


def load_hyperparams(path) -> _Params:
    hp = load_yaml(path)
    if not isinstance(hp, dict):
        raise ValueError(f"{path}: expected a flat mapping of hyper-parameters")
    p = _Params({k: (None if v == 'None' else v) for k, v in hp.items()})
    p.setdefault("nettype", "swin")
    if "n_in_channels" not in p:                      # derived in train.py:88-98, dumped with the rest when training ran
        n = len(p["in_channels"])
        p["n_in_channels"] = n + int(bool(p.get("add_zenith", False))) + 2 * int(bool(p.get("add_landmask", False))) + \
            int(bool(p.get("add_orography", False)))
        p["n_out_channels"] = len(p["out_channels"])
    for k, d in (("activation_ckpt", False), ("residual", False), ("add_orography", False), ("add_landmask", False),
                 ("add_zenith", False), ("mlp_ratio", 4), ("full_pos_embed", True), ("rel_pos", False), ("drop_path_rate", 0.0)):
        p.setdefault(k, d)
    return p


def load_model_state(model: torch.nn.Module, state: dict) -> None:
    """load a reference state_dict whatever wrapper prefixes it carries (train.py:381-389 strips 'module.' by position)"""
    want = set(model.state_dict().keys())
    for strip in ("", "module.", "module.model.", "model."):
        cand = {(k[len(strip):] if k.startswith(strip) else k): v for k, v in state.items()}
        for add in ("", "model."):
            cand2 = {add + k: v for k, v in cand.items()}
            if set(cand2.keys()) == want:
                model.load_state_dict(cand2, strict=True)
                return
    missing = sorted(want - set(state.keys()))[:5]
    raise KeyError(f"checkpoint keys do not match the model under any known prefix (first missing: {missing})")


def load_registry_model(model_dir: str, device="cuda:0", trust_checkpoint: bool = False):
    """-> (single-step model in eval mode on `device`, params, (means, stds) or None).
    `weights.tar` is a published / third-party file: it is read with torch's restricted unpickler (tensors, state dicts and
    plain containers only).  trust_checkpoint=True falls back to the full unpickler -- arbitrary code execution from the file --
    for checkpoints that carry other objects; only for files you produced yourself."""
    p = load_hyperparams(os.path.join(model_dir, "hyperparams.yaml"))
    p["n_future"] = 0
    model = get_model(p)
    ck = torch.load(os.path.join(model_dir, "weights.tar"), map_location="cpu", weights_only=not trust_checkpoint)
    load_model_state(model, ck["model_state"] if "model_state" in ck else ck)
    stats = None
    gm, gs = os.path.join(model_dir, "global_means.npy"), os.path.join(model_dir, "global_stds.npy")
    if os.path.isfile(gm) and os.path.isfile(gs):
        stats = (np.load(gm), np.load(gs))
    return model.to(device).eval(), p, stats


@torch.no_grad()
def rollout(model, x0: torch.Tensor, n_steps: int, coszen: torch.Tensor | None = None, n_invar: int = 0) -> torch.Tensor:
    """x0 [B, Cin, H, W] (fields | zenith(t0) | invariants) -> forecasts [B, n_steps, Cout, H, W].
    coszen [B, n_steps - 1, H, W]: cos-zenith of the steps 1 .. n_steps - 1 inputs (required iff the model takes one)."""
    net = model.model if hasattr(model, "model") else model
    B, _, H, W = x0.shape
    Cout = net.out_chans
    out = torch.empty(B, n_steps * Cout, H, W, dtype=torch.float32, device=x0.device)
    invars = x0[:, x0.shape[1] - n_invar:] if n_invar else None
    x = x0
    for s in range(n_steps):
        extra = None
        if s + 1 < n_steps:
            parts = ([coszen[:, s:s + 1]] if coszen is not None else []) + ([invars] if n_invar else [])
            extra = torch.cat(parts, 1).float() if parts else x0.new_empty(B, 0, H, W)
        _, x = net.forward_rollout(x, out, s * Cout, extra)
    return out.view(B, n_steps, Cout, H, W)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--registry", required=True, help="model folder with hyperparams.yaml + weights.tar")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--init", default=None, help=".npy [B, Cin, H, W] normalised initial condition (default: seeded N(0,1))")
    ap.add_argument("--out", default=None)
    ap.add_argument("--trust-checkpoint", action="store_true", help="read weights.tar with the full unpickler (runs code from the file)")
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    model, p, _ = load_registry_model(a.registry, dev, trust_checkpoint=a.trust_checkpoint)
    H, W = p["img_size"]
    n_invar = 2 * int(bool(p["add_landmask"])) + int(bool(p["add_orography"]))
    if a.init:
        x0 = torch.from_numpy(np.load(a.init)).float().to(dev)
    else:
        x0 = torch.randn(1, p["n_in_channels"], H, W, generator=torch.Generator().manual_seed(0)).to(dev)
    cz = None
    if p["add_zenith"]:
        from .utils.data_loader_era5 import cos_zenith
        cz = torch.stack([cos_zenith(2018, 6.0 * (s + 1), H, W) for s in range(a.steps - 1)], 0).unsqueeze(0).expand(x0.shape[0], -1, -1, -1).to(dev) \
            if a.steps > 1 else None
    y = rollout(model, x0, a.steps, cz, n_invar)
    print(f"forecast {tuple(y.shape)}: per-step rms " + " ".join(f"{float(y[:, s].square().mean().sqrt()):.4f}" for s in range(a.steps)))
    if a.out:
        np.save(a.out, y.cpu().numpy())


if __name__ == "__main__":
    main()
