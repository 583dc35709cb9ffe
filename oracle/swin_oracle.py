"""CPU oracle for the SwinV2 weather hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU, fp32 or fp64)
restatement of the arithmetic of the reference model
(`/root/reference/networks/swinv2_global.py`, `networks/helpers.py`,
`utils/losses.py`, `utils/grids.py`).  It is the *checker* for the HIP kernels:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product package `swin_v2_weather_amd` never
imports anything under `oracle/`.

It is written from the semantic spec, not from the
reference's module code: windows are addressed by explicit index math
(roll + partition folded into one gather), the shift mask is the closed form,
and everything is a pure function of a flat `{name: tensor}` dict that uses
the reference's state_dict key names (so reference checkpoints / fixtures load
without renaming).

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real
reference (with shims for the absent `timm.layers` / `ruamel.yaml`) in the
build container, and `tests/test_oracle_golden.py` checks every function here
against those committed vectors.  The `timm.layers.Mlp/DropPath` semantics are
third-party (timm, version unpinned by the reference; header says v0.9.2) and
are restated here from timm's published behaviour:
    Mlp      = fc1 -> act -> Dropout(p0) -> fc2 -> Dropout(p1)
    DropPath = x * bernoulli(1-p)/(1-p), mask shape (B,1,...,1), train only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# Optional rounding emulation (tests only): when set to a callable, it is applied at exactly the points where the HIP
# path stores a value as bf16 (GEMM operands, normalised q/k, v, softmax probabilities, branch outputs).  The default
# (None) is the exact fp32/fp64 restatement of the reference.  Gradients pass straight through the rounding.
_ROUND = None
_SOFTMAX = "row_max"


def set_rounding(fn, softmax: str = "row_max") -> None:
    """fn: rounding applied at the HIP path's bf16 storage points (None: exact arithmetic).  softmax: the forward softmax regime of
    the rounding mode, DECLARED by the caller (the tests state it per geometry and check the library's kernel choice against it with
    swv2_attn_fwd_regime; the oracle never infers it from shapes):
      "row_max"        exponent reference = the row maximum, normaliser = sum of the exact exponentials (csrc/attn.hip forward);
      "operand_folded" csrc/attn2.hip (attn_fwd3_kernel, attn_fwd3w_kernel, attn_fwd3b_kernel): the scale enters the QK^T product inside
                       the query operand, sigma log2(e) q^ as TWO bf16 parts hi + lo (hi = bf16(x), lo = bf16(x - hi): the scaled query is
                       exact to ~2^-17 instead of fp32); reference = sigma (+ the largest entry of the head's CPB table, which enters the
                       scores through an identity MFMA) while 2 sigma log2(e) + (max - min of the table, log2 domain) <= 80 in windows
                       without a shift mask (cosines are bounded), the row maximum otherwise; normaliser = sum of the bf16-ROUNDED
                       exponentials (an all-ones MFMA operand)."""
    global _ROUND, _SOFTMAX
    assert softmax in ("row_max", "operand_folded")
    _ROUND = fn
    _SOFTMAX = softmax if fn is not None else "row_max"


def bf16_round(x: Tensor) -> Tensor:
    """straight-through bf16 rounding: value of x.bfloat16(), gradient of identity"""
    return x + (x.detach().to(torch.bfloat16).to(x.dtype) - x.detach())


def _r(x: Tensor) -> Tensor:
    return x if _ROUND is None else _ROUND(x)


class _GradRound(torch.autograd.Function):
    """identity whose backward rounds the gradient like the forward rounding in force when it was applied"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.fn(g.detach()), None


def _g(x: Tensor) -> Tensor:
    """Marks a GEMM OUTPUT: in rounding mode the gradient that flows back into it is rounded too -- the HIP path hands every
    backward product its dY operand as bf16 (d(qkv), d(a1), d(h), d(a2) are stored as bf16 by the kernel in front; the head
    and PatchEmbed gradients are converted when they are loaded)."""
    return x if _ROUND is None else _GradRound.apply(x, _ROUND)


LN_EPS = 1e-5               # torch.nn.LayerNorm default (swinv2_global.py:376,387 use nn.LayerNorm)
LOGIT_MAX = math.log(100.0)  # swinv2_global.py:305  clamp(max=log(1/0.01))


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass
class SwinCfg:
    """Mirror of the ctor mapping in swinv2_global.py:57-74 / :683-775."""
    img_size: Tuple[int, int]
    patch_size: int
    depth: int
    num_heads: int
    in_chans: int
    out_chans: int
    embed_dim: int
    window_ratio: int
    mlp_ratio: float = 4.0
    drop_path_rate: float = 0.0
    full_pos_embed: bool = True
    rel_pos: bool = False
    residual: bool = False
    meta_hidden: int = 384          # swinv2_global.py:225
    meta_dropout: float = 0.125     # swinv2_global.py:245

    @property
    def grid(self) -> Tuple[int, int]:
        return (self.img_size[0] // self.patch_size, self.img_size[1] // self.patch_size)

    @property
    def window(self) -> Tuple[int, int]:
        # swinv2_global.py:714-715 window = img_size // ratio, then clamped to the grid (:398-401)
        w = (self.img_size[0] // self.window_ratio, self.img_size[1] // self.window_ratio)
        g = self.grid
        return (min(w[0], g[0]), min(w[1], g[1]))

    def shift(self, block_index: int) -> Tuple[int, int]:
        # swinv2_global.py:612 odd blocks shift by window//2; :398-401 no shift when window >= grid
        if block_index % 2 == 0:
            return (0, 0)
        tw = (self.img_size[0] // self.window_ratio, self.img_size[1] // self.window_ratio)
        g = self.grid
        w = self.window
        return tuple(0 if g[i] <= w[i] else tw[i] // 2 for i in range(2))  # type: ignore

    def drop_path(self, block_index: int) -> float:
        # swinv2_global.py:737 linspace(0, rate, depth)
        if self.depth == 1:
            return 0.0
        return float(torch.linspace(0, self.drop_path_rate, self.depth)[block_index])

    @staticmethod
    def from_params(params) -> "SwinCfg":
        return SwinCfg(img_size=tuple(params.img_size), patch_size=params.patch_size, depth=params.depth,
                       num_heads=params.num_heads, in_chans=params.n_in_channels,
                       out_chans=params.n_out_channels, embed_dim=params.embed_dim,
                       window_ratio=params.window_ratio, mlp_ratio=params.mlp_ratio,
                       drop_path_rate=params.drop_path_rate, full_pos_embed=params.full_pos_embed,
                       rel_pos=params.rel_pos, residual=params.residual)


# --------------------------------------------------------------------------
# addressing: roll + window partition as one gather (swinv2_global.py:457,89-101,105-119,476)
# --------------------------------------------------------------------------
def window_token_index(gh: int, gw: int, wh: int, ww: int, sh: int, sw: int) -> Tensor:
    """Flat source index (i*gw + j) of every (window, token): shape [nW, L].

    x_win[b*nW + wi*(gw/ww) + wj, r*ww + c] = x[b, (wi*wh + r + sh) % gh, (wj*ww + c + sw) % gw]
    """
    assert gh % wh == 0 and gw % ww == 0, "window must divide the patch grid (swinv2_global.py:99)"
    wi = torch.arange(gh // wh).view(-1, 1, 1, 1)
    wj = torch.arange(gw // ww).view(1, -1, 1, 1)
    r = torch.arange(wh).view(1, 1, -1, 1)
    c = torch.arange(ww).view(1, 1, 1, -1)
    i = (wi * wh + r + sh) % gh
    j = (wj * ww + c + sw) % gw
    return (i * gw + j).reshape((gh // wh) * (gw // ww), wh * ww)


def roll_partition(x: Tensor, wh: int, ww: int, sh: int, sw: int) -> Tensor:
    """[B, gh, gw, C] -> [B*nW, L, C]."""
    B, gh, gw, C = x.shape
    idx = window_token_index(gh, gw, wh, ww, sh, sw).reshape(-1)
    return x.reshape(B, gh * gw, C)[:, idx, :].reshape(-1, wh * ww, C)


def reverse_unroll(xw: Tensor, gh: int, gw: int, wh: int, ww: int, sh: int, sw: int) -> Tensor:
    """[B*nW, L, C] -> [B, gh, gw, C]; exact inverse permutation of roll_partition."""
    C = xw.shape[-1]
    idx = window_token_index(gh, gw, wh, ww, sh, sw).reshape(-1)
    B = xw.shape[0] * xw.shape[1] // (gh * gw)
    out = torch.empty(B, gh * gw, C, dtype=xw.dtype)
    out[:, idx, :] = xw.reshape(B, gh * gw, C)
    return out.reshape(B, gh, gw, C)


def shift_mask(gh: int, gw: int, wh: int, ww: int, sh: int, sw: int) -> Optional[Tensor]:
    """Closed form of swinv2_global.py:403-424: [nW, L, L] in {0, -100}, None when unshifted.

    In rolled coordinates row i' has region id 1 iff i' >= gh - sh (the second slice
    `slice(-shift, None)`); the first slice writes id 0; rows in between keep the
    zero initialisation.  No bands along W (periodic longitude).
    """
    if sh == 0 and sw == 0:
        return None
    nwh, nww = gh // wh, gw // ww
    rows = (torch.arange(nwh).view(-1, 1) * wh + torch.arange(wh).view(1, -1))   # [nwh, wh] rolled row
    rid = (rows >= gh - sh).to(torch.float32) if sh > 0 else torch.zeros(nwh, wh)
    tok = rid.view(nwh, 1, wh, 1).expand(nwh, nww, wh, ww).reshape(nwh * nww, wh * ww)
    diff = tok.unsqueeze(1) - tok.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def rel_coords_log(wh: int, ww: int) -> Tensor:
    """swinv2_global.py:251-261 -> [L*L, 2], row index t_q*L + t_k, value sign(d)*log(1+|d|)."""
    r = torch.arange(wh).view(-1, 1).expand(wh, ww).reshape(-1)
    c = torch.arange(ww).view(1, -1).expand(wh, ww).reshape(-1)
    d = torch.stack([r.view(-1, 1) - r.view(1, -1), c.view(-1, 1) - c.view(1, -1)], dim=-1)
    d = d.reshape(-1, 2).to(torch.float32)
    return torch.sign(d) * torch.log1p(d.abs())


# --------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def cpb_bias(p: Dict[str, Tensor], pre: str, wh: int, ww: int, heads: int,
             training: bool, drop_p: float = 0.125) -> Tensor:
    """Continuous position bias [h, L, L] (swinv2_global.py:274-287): meta MLP on log-coords.

    Dropout(0.125) on the hidden layer is active in training mode (:245); it is drawn with
    F.dropout so the torch RNG stream is consumed exactly like the reference's nn.Dropout.
    """
    L = wh * ww
    R = rel_coords_log(wh, ww).to(p[pre + "meta_mlp.fc1.weight"].dtype)
    hdn = torch.relu(R @ p[pre + "meta_mlp.fc1.weight"].T + p[pre + "meta_mlp.fc1.bias"])
    hdn = F.dropout(hdn, drop_p, training)
    o = hdn @ p[pre + "meta_mlp.fc2.weight"].T + p[pre + "meta_mlp.fc2.bias"]     # [L*L, h]
    o = o.T.reshape(heads, L, L)
    if _ROUND is not None:      # the kernels hold the bias table in the log2 domain as bf16
        o = _r(o * 1.4426950408889634) / 1.4426950408889634
    return o


def attention_core(qkv: Tensor, logit_scale: Tensor, heads: int,
                   bias: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """Cosine window attention on already-partitioned windows (swinv2_global.py:298-318).

    qkv [Bw, L, 3C] with feature index s*C + head*d + j; returns [Bw, L, C] (heads concatenated).
    """
    Bw, L, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    q, k, v = qkv.reshape(Bw, L, 3, heads, d).permute(2, 0, 3, 1, 4)        # each [Bw, h, L, d]
    qn = _r(q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12))
    kn = _r(k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12))
    v = _r(v)
    return attention_core_normed(qn, kn, v, logit_scale, bias, mask)


def attention_core_normed(qn: Tensor, kn: Tensor, v: Tensor, logit_scale: Tensor,
                          bias: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """attention_core from the L2-normalised q, k and v, each [Bw, h, L, d] (what the attention kernels are handed)"""
    Bw, heads, L, d = qn.shape
    C = heads * d
    S = torch.einsum("bhqd,bhkd->bhqk", qn, kn)
    S = S * torch.exp(torch.clamp(logit_scale, max=LOGIT_MAX)).view(1, heads, 1, 1)
    if bias is not None:
        S = S + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        S = (S.reshape(Bw // nW, nW, heads, L, L) + mask.view(1, nW, 1, L, L)).reshape(Bw, heads, L, L)
    if _ROUND is None:
        P = torch.softmax(S, dim=-1)
        return torch.einsum("bhqk,bhkd->bqhd", P, v).reshape(Bw, L, C)
    # HIP path: un-normalised exp rounded to bf16 for the P.V product, one division at the end; the backward follows the
    # kernels' data flow as well (_AttnCoreEmu), so the tests can hold d logit_scale -- sigma sum(dS cos), a sum with heavy
    # cancellation -- to a tight bar instead of the 12 - 15 % that exact autograd of this forward leaves.
    rowmax = S.detach().max(dim=-1, keepdim=True).values
    ref, rounded_sum = rowmax, False
    if _SOFTMAX == "operand_folded":          # declared by the caller (set_rounding), never inferred from the shapes
        sig = torch.exp(torch.clamp(logit_scale.detach(), max=LOGIT_MAX))
        l2e = math.log2(math.e)
        if bias is not None:                  # (the table as the kernels hold it: bf16 in the log2 domain)
            bmax, bmin = bias.detach().flatten(1).max(1).values, bias.detach().flatten(1).min(1).values
        else:
            bmax = bmin = torch.zeros_like(sig)
        fixed = (2.0 * sig * l2e + (bmax - bmin) * l2e <= 80.0).view(1, heads, 1, 1)
        if mask is not None:
            nW = mask.shape[0]
            masked_w = (mask != 0).flatten(1).any(1).repeat(Bw // nW).view(Bw, 1, 1, 1)
            fixed = fixed & ~masked_w
        ref = torch.where(fixed, (sig + bmax).view(1, heads, 1, 1).to(S.dtype).expand_as(rowmax), rowmax)
        rounded_sum = True
    sigma = torch.exp(torch.clamp(logit_scale, max=LOGIT_MAX))
    mfull = None
    if mask is not None:
        nW = mask.shape[0]
        mfull = mask.view(1, nW, 1, L, L).expand(Bw // nW, nW, 1, L, L).reshape(Bw, 1, L, L)
    O = _AttnCoreEmu.apply(qn, kn, v, sigma, bias, mfull, ref.detach(), rounded_sum, rounded_sum)      # (both regimes with a rounded sum fold the scale)
    return O.permute(0, 2, 1, 3).reshape(Bw, L, C)


class _AttnCoreEmu(torch.autograd.Function):
    """Forward AND backward of the attention core in the kernels' rounding mode (csrc/attn.hip, attn2.hip; tests only).
    Forward: E = exp(S - ref), bf16(E) feeds P.V, one division by the row sum (of the rounded or the exact E), O stored as
    bf16, lse = ref + ln(sum).  Backward (flash-style recompute, as attn_bwd_kernel): P = exp(S - lse) in fp32,
    delta = rowsum(dO * O) with the STORED bf16 O, dP = dO v^T, dS = P (dP - delta); dV from bf16(P), dq / dk / d sigma from
    bf16(dS), d bias from the un-rounded dS."""

    @staticmethod
    def forward(ctx, qn, kn, v, sigma, bias, mask, ref, rounded_sum, folded_scale=False):
        if folded_scale:
            # sigma log2(e) q^ as the kernel's two-part operand: hi = round(x), lo = round(x - hi) (the fp32 rounding of the product
            # itself, 2^-24, is not modelled; with an identity rounding function hi = x, lo = 0: the exact logits)
            x = qn * (sigma * math.log2(math.e)).view(1, -1, 1, 1)
            hi = _r(x)
            lo = _r(x - hi)
            S = torch.einsum("bhqd,bhkd->bhqk", hi + lo, kn) / math.log2(math.e)
        else:
            S = torch.einsum("bhqd,bhkd->bhqk", qn, kn) * sigma.view(1, -1, 1, 1)
        if bias is not None:
            S = S + bias.unsqueeze(0)
        if mask is not None:
            S = S + mask
        E = torch.exp(S - ref)
        Er = _r(E)
        ssum = (Er if rounded_sum else E).sum(-1, keepdim=True)
        O = _r(torch.einsum("bhqk,bhkd->bhqd", Er, v) / ssum)
        ctx.save_for_backward(qn, kn, v, sigma, bias, mask, O, ref + torch.log(ssum))
        return O

    @staticmethod
    def backward(ctx, dO):
        qn, kn, v, sigma, bias, mask, O, lse = ctx.saved_tensors
        dO = _r(dO)                                   # d(oh) arrives as bf16 from the proj backward
        cos = torch.einsum("bhqd,bhkd->bhqk", qn, kn)
        S = cos * sigma.view(1, -1, 1, 1)
        if bias is not None:
            S = S + bias.unsqueeze(0)
        if mask is not None:
            S = S + mask
        P = torch.exp(S - lse)
        delta = (dO * O).sum(-1, keepdim=True)
        dP = torch.einsum("bhqd,bhkd->bhqk", dO, v)
        dS = P * (dP - delta)
        dV = torch.einsum("bhqk,bhqd->bhkd", _r(P), dO)
        dSr = _r(dS)
        sg = sigma.view(1, -1, 1, 1)
        dqn = sg * torch.einsum("bhqk,bhkd->bhqd", dSr, kn)
        dkn = sg * torch.einsum("bhqk,bhqd->bhkd", dSr, qn)
        dsigma = (dSr * cos).sum((0, 2, 3))
        dbias = dS.sum(0) if bias is not None else None
        return dqn, dkn, dV, dsigma, dbias, None, None, None, None


def window_attention(xw: Tensor, p: Dict[str, Tensor], pre: str, heads: int,
                     bias: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """qkv Linear -> attention_core -> proj Linear on [Bw, L, C] windows."""
    qkv = _g(_r(xw) @ _r(p[pre + "qkv.weight"]).T + p[pre + "qkv.bias"])
    o = attention_core(qkv, p[pre + "logit_scale"], heads, bias, mask)
    return _r(_g(o @ _r(p[pre + "proj.weight"]).T + p[pre + "proj.bias"]))


def drop_path_scale(B: int, prob: float, training: bool, like: Tensor) -> Optional[Tensor]:
    """timm DropPath: mask (B,1,...) = bernoulli(keep)/keep; None when inactive."""
    if prob == 0.0 or not training:
        return None
    keep = 1.0 - prob
    m = like.new_empty((B,)).bernoulli_(keep)
    if keep > 0.0:
        m.div_(keep)
    return m


def block_forward(x: Tensor, p: Dict[str, Tensor], pre: str, cfg: SwinCfg, index: int,
                  training: bool = False, bias_override: Optional[Tensor] = None,
                  dp_override: Optional[Tuple[Optional[Tensor], Optional[Tensor]]] = None) -> Tensor:
    """One post-norm Swin block on [B, gh, gw, C] (swinv2_global.py:480-497).  `bias_override` / `dp_override` replace
    the stochastic pieces (CPB table drawn with dropout, the two DropPath scale vectors) by given tensors, so a run
    whose random draws happened elsewhere (on the GPU) can be replayed exactly."""
    B, gh, gw, C = x.shape
    wh, ww = cfg.window
    sh, sw = cfg.shift(index)
    xw = roll_partition(x, wh, ww, sh, sw)
    bias = cpb_bias(p, pre + "attn.", wh, ww, cfg.num_heads, training, cfg.meta_dropout) if cfg.rel_pos else None
    if bias_override is not None:
        bias = bias_override
        if _ROUND is not None:
            bias = _r(bias * 1.4426950408889634) / 1.4426950408889634
    mask = shift_mask(gh, gw, wh, ww, sh, sw)
    if mask is not None:
        mask = mask.to(x.dtype)
    a = window_attention(xw, p, pre + "attn.", cfg.num_heads, bias, mask)
    a = reverse_unroll(a, gh, gw, wh, ww, sh, sw)
    a = layer_norm(a, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
    dp = cfg.drop_path(index)
    s1 = drop_path_scale(B, dp, training, x) if dp_override is None else dp_override[0]
    x = x + (a if s1 is None else a * s1.view(B, 1, 1, 1))
    m = _r(gelu_erf(_r(_g(_r(x) @ _r(p[pre + "mlp.fc1.weight"]).T + p[pre + "mlp.fc1.bias"]))))
    m = _r(_g(m @ _r(p[pre + "mlp.fc2.weight"]).T + p[pre + "mlp.fc2.bias"]))
    m = layer_norm(m, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
    s2 = drop_path_scale(B, dp, training, x) if dp_override is None else dp_override[1]
    return x + (m if s2 is None else m * s2.view(B, 1, 1, 1))


def patch_embed(x: Tensor, p: Dict[str, Tensor], pre: str, P: int) -> Tensor:
    """[B,Cin,H,W] -> [B,gh,gw,C]: k=s=P conv as a per-patch GEMM, + bias, LayerNorm (swinv2_global.py:537-546)."""
    B, Cin, H, W = x.shape
    gh, gw = H // P, W // P
    w = p[pre + "proj.weight"]                                   # [C, Cin, P, P]
    patches = x.reshape(B, Cin, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh, gw, Cin * P * P)
    e = _r(_g(_r(patches) @ _r(w.reshape(w.shape[0], -1)).T + p[pre + "proj.bias"]))
    return layer_norm(e, p[pre + "norm.weight"], p[pre + "norm.bias"])


def patch_merging(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """[B,H,W,C] -> [B,H/2,W/2,2C]; channel order (w-parity, h-parity, C) (swinv2_global.py:519-523)."""
    B, H, W, C = x.shape
    parts = [x[:, hp::2, wp::2, :] for wp in (0, 1) for hp in (0, 1)]
    m = torch.cat(parts, dim=-1)
    m = layer_norm(m, p[pre + "norm.weight"], p[pre + "norm.bias"])
    return _g(_r(m) @ _r(p[pre + "reduction.weight"]).T)


def head_unpatchify(e: Tensor, w_head: Tensor, P: int, out_chans: int,
                    skip: Optional[Tensor]) -> Tensor:
    """[B,gh,gw,C] -> [B,Cout,H,W]: y[b,c,P*i+p,P*j+q] = (e W^T)[b,i,j,(p*P+q)*Cout+c] (+skip) (:784-802)."""
    B, gh, gw, C = e.shape
    z = _g(_r(e) @ _r(w_head).T).reshape(B, gh, gw, P, P, out_chans)
    y = z.permute(0, 5, 1, 3, 2, 4).reshape(B, out_chans, gh * P, gw * P)
    if skip is not None:
        y = y + skip[:, :out_chans]
    return y


def model_forward(x: Tensor, p: Dict[str, Tensor], cfg: SwinCfg, training: bool = False,
                  prefix: str = "") -> Tensor:
    """Whole network, swinv2_global.py:794-803."""
    e = patch_embed(x, p, prefix + "patch_embed.", cfg.patch_size)
    if cfg.full_pos_embed:
        e = e + p[prefix + "pos_embed"].permute(0, 2, 3, 1)
    for i in range(cfg.depth):
        e = block_forward(e, p, f"{prefix}stages.0.blocks.{i}.", cfg, i, training)
    return head_unpatchify(e, p[prefix + "head.weight"], cfg.patch_size, cfg.out_chans,
                           x if cfg.residual else None)


def multistep_forward(inp: Tensor, coszen: Optional[Tensor], p: Dict[str, Tensor], cfg: SwinCfg,
                      n_future: int, n_invar: int, training: bool = False, prefix: str = "model.") -> Tensor:
    """Autoregressive rollout, helpers.py:26-41."""
    outs = []
    cur = inp
    invars = inp[:, -n_invar:] if n_invar else None
    for step in range(n_future + 1):
        pred = model_forward(cur, p, cfg, training, prefix)
        outs.append(pred)
        if step == n_future:
            break
        cur = pred
        if coszen is not None:
            cur = torch.cat([cur, coszen[:, step:step + 1]], dim=1)
        if invars is not None:
            cur = torch.cat([cur, invars], dim=1)
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------
# loss (losses.py:30-232, grids.py:62-117) -- 'l2' family with naive quadrature
# --------------------------------------------------------------------------
def quadrature_weights(H: int, W: int, dtype=torch.float32) -> Tensor:
    """grids.py:68-76,93-94: sin(linspace(0,pi,H)) latitude weights, normalised to sum 1 -> [H, W]."""
    jac = torch.clamp(torch.sin(torch.linspace(0, math.pi, H, dtype=dtype)), min=0.0)
    dA = (2 * math.pi / W) * (math.pi / H)
    q = (dA * jac).unsqueeze(1).repeat(1, W)
    q = q * (4.0 * math.pi) / q.sum()
    return q / (4.0 * math.pi)


def auto_channel_weights(channel_names: Sequence[str], n_out: int) -> Tensor:
    """losses.py:57-68."""
    w = torch.ones(n_out, dtype=torch.float32)
    for c, chn in enumerate(channel_names):
        if chn in ("u10m", "v10m", "u100m", "v100m", "tp", "sp", "msl", "tcwv"):
            w[c] = 0.1
        elif chn in ("t2m", "2d"):
            w[c] = 1.0
        elif chn[0] in "zuvtrq":
            w[c] = 0.001 * float(chn[1:])
        else:
            w[c] = 0.01
    return w


def loss_channel_weights(loss: str, n_out: int, n_future: int, channel_names=None,
                         channel_weights="none", global_stds: Optional[Tensor] = None,
                         time_diff_stds: Optional[Tensor] = None, dt: int = 1,
                         training: bool = True) -> Tensor:
    """Per-channel weights [1, (n_future+1)*n_out] exactly as LossHandler builds them (losses.py:47-150)."""
    flags = set(loss.split())
    if "weighted" in flags:
        if channel_weights == "auto":
            cw = auto_channel_weights(channel_names, n_out)
        else:
            cw = torch.tensor(channel_weights, dtype=torch.float32)
    else:
        cw = torch.ones(n_out, dtype=torch.float32)
    cw = cw.reshape(1, -1, 1, 1)
    cw = cw / cw.sum()
    if "temp-std" in flags:
        tvw = global_stds.reshape(1, -1, 1, 1) / (math.sqrt(dt) * time_diff_stds.reshape(1, -1, 1, 1) + 1e-6)
        if "squared" in flags:
            tvw = tvw ** 2
        cw = cw * tvw
    if training:
        ms = torch.ones(n_future + 1, dtype=torch.float32) / float(n_future + 1)
        cw = cw * ms.reshape(-1, 1, 1, 1)          # broadcast -> [n_future+1, n_out, 1, 1]
    return cw.reshape(1, -1)


def geometric_l2_loss(prd: Tensor, tar: Tensor, chw: Tensor, loss: str) -> Tensor:
    """GeometricLpLoss p=2 (losses.py:188-232) with the naive quadrature.  Sum over batch and channels.

    Note quirk 9: plain 'l2' ignores jacobian='flat' and is sphere-weighted as well;
    and without the word 'squared' the per-channel norm takes a square root; without 'absolute'
    the relative form divides by the target norm.
    """
    flags = set(loss.split())
    B, Ctot, H, W = prd.shape
    q = quadrature_weights(H, W, prd.dtype)
    squared = ("squared" in flags) and ("geometric" in flags)   # plain 'l2' never passes squared (losses.py:112-113)
    dn = ((prd - tar).abs() ** 2 * q).sum(dim=(-2, -1)).reshape(B, -1)
    if "absolute" in flags:
        n = dn
    else:
        tn = (tar.abs() ** 2 * q).sum(dim=(-2, -1)).reshape(B, -1)
        n = dn / tn
    if not squared:
        n = n ** 0.5
    return (chw * n).sum()


# --------------------------------------------------------------------------
# convenience: an nn.Module view of the oracle (used for loss-curve fixtures and the CPU baseline)
# --------------------------------------------------------------------------
class OracleNet(torch.nn.Module):
    """Holds a flat parameter dict under the reference's state_dict names."""

    def __init__(self, cfg: SwinCfg, state: Dict[str, Tensor], n_future: int = 0, n_invar: int = 0):
        super().__init__()
        self.cfg, self.n_future, self.n_invar = cfg, n_future, n_invar
        self.keys = list(state.keys())
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(state[k].detach().clone()) for k in self.keys])

    def pdict(self) -> Dict[str, Tensor]:
        return {k: v for k, v in zip(self.keys, self.plist)}

    def forward(self, inp: Tensor, coszen: Optional[Tensor] = None) -> Tensor:
        return multistep_forward(inp, coszen, self.pdict(), self.cfg, self.n_future, self.n_invar,
                                 self.training, prefix="model.")
