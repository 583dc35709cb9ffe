"""Import shims that let the read-only reference (/root/reference) be imported in the build
container, where `timm`, `ruamel.yaml`, `torch_harmonics` ... are not installed.

Only used by `make_golden.py` (fixture generation in the build container, where /root/reference exists); nothing under
`tests/test_*.py` imports it, so no test reads /root/reference at run time.

The `timm.layers` stand-ins are our restatement of timm's published semantics
(timm >= 0.9): see oracle/swin_oracle.py header.  They are NOT part of the product.
"""
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


class _Mlp(nn.Module):
    """timm.layers.Mlp: fc1 -> act -> drop1 -> (norm=Identity) -> fc2 -> drop2, bias on both."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        drops = drop if isinstance(drop, (tuple, list)) else (drop, drop)
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drops[0])
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drops[1])

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class _DropPath(nn.Module):
    """timm.layers.DropPath (stochastic depth per sample, scale_by_keep=True)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        m = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            m.div_(keep)
        return x * m


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _assert(cond, msg):
    assert cond, msg


def install():
    """Register stand-in modules and put the reference on sys.path."""
    if "timm.layers" not in sys.modules:
        timm = types.ModuleType("timm")
        layers = types.ModuleType("timm.layers")
        layers.Mlp, layers.DropPath = _Mlp, _DropPath
        layers.ClassifierHead = type("ClassifierHead", (nn.Module,), {})
        layers.to_2tuple, layers._assert = _to_2tuple, _assert
        timm.layers = layers
        sys.modules["timm"], sys.modules["timm.layers"] = timm, layers
    if "ruamel.yaml" not in sys.modules:
        ruamel = types.ModuleType("ruamel")
        ryaml = types.ModuleType("ruamel.yaml")
        ryaml.YAML = type("YAML", (), {})
        ruamel.yaml = ryaml
        sys.modules["ruamel"], sys.modules["ruamel.yaml"] = ruamel, ryaml
    if "torch_harmonics" not in sys.modules:
        th = types.ModuleType("torch_harmonics")
        thq = types.ModuleType("torch_harmonics.quadrature")
        thq.legendre_gauss_weights = thq.clenshaw_curtiss_weights = None
        th.quadrature = thq
        th.RealSHT = th.RealVectorSHT = None
        sys.modules["torch_harmonics"], sys.modules["torch_harmonics.quadrature"] = th, thq
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def import_reference():
    """Returns (swinv2_global module, helpers module, losses module)."""
    install()
    # the reference packages are called `networks` / `utils`; make sure ours do not shadow them here
    for name in [m for m in sys.modules if m == "networks" or m.startswith("networks.") or
                 m == "utils" or m.startswith("utils.")]:
        del sys.modules[name]
    import importlib
    sw = importlib.import_module("networks.swinv2_global")
    hp = importlib.import_module("networks.helpers")
    try:
        ls = importlib.import_module("utils.losses")
    except Exception:       # pragma: no cover - losses need extra shims on some images
        ls = None
    return sw, hp, ls
