"""LossHandler with the reference's surface and semantics (utils/losses.py:30-232) on the fused quadrature kernels.

The O(B*C*H*W) work -- the latitude-weighted sums of (prd - tar)^2 and tar^2 per (sample, channel) plane and the
gradient d loss / d prd -- is two HIP kernels (swv2_loss_sums / swv2_loss_grad); the remaining arithmetic is on
[B, C] tensors.  Supported: the 'l2' family used by every entry of config/swin.yaml ('l2', 'squared geometric l2',
'weighted absolute|relative temp-std squared geometric l2').  l1 / geometric h1 are not selected by any config.
"""
import math

import os

import numpy as np
import torch
from torch import nn

from .. import ops
from .grids import naive_quadrature_weights


class _QuadSums(torch.autograd.Function):
    """sums[b, c] = (sum_hw q[h] (prd - tar)^2, sum_hw q[h] tar^2); backward: dprd = 2 dS0[b,c] q[h] (prd - tar)"""

    @staticmethod
    def forward(ctx, prd, tar, qw):
        prd, tar = prd.contiguous().float(), tar.contiguous().float()
        B, C = prd.shape[:2]
        sums = torch.zeros(B, C, 2, dtype=torch.float32, device=prd.device)
        ops.loss_sums(prd, tar, qw, sums)
        ctx.save_for_backward(prd, tar, qw)
        return sums

    @staticmethod
    def backward(ctx, dsums):
        prd, tar, qw = ctx.saved_tensors
        coef = (2.0 * dsums[..., 0]).contiguous().float()
        dprd = torch.empty_like(prd)
        ops.loss_grad(prd, tar, qw, coef, dprd)
        return dprd, None, None


class _GeoL2(torch.autograd.Function):
    """The whole LossHandler value as one node: loss_sums (one pass over prd / tar), loss_finalize (the [B, C] arithmetic +
    the backward coefficients), and in backward one loss_grad pass.  Same value and gradient as _QuadSums + torch ops."""

    @staticmethod
    def forward(ctx, prd, tar, qw, chw, absolute, squared):
        prd, tar = prd.contiguous().float(), tar.contiguous().float()
        B, C = prd.shape[:2]
        buf = torch.zeros(B * C * 3 + 1, dtype=torch.float32, device=prd.device)      # sums | coef | loss: one memset
        sums, coef, loss = buf[:B * C * 2].view(B, C, 2), buf[B * C * 2:B * C * 3], buf[B * C * 3:]
        ops.loss_sums(prd, tar, qw, sums)
        ops.loss_finalize(sums, chw, absolute, squared, loss, coef)
        ctx.save_for_backward(prd, tar, qw, coef)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        prd, tar, qw, coef = ctx.saved_tensors
        dprd = torch.empty_like(prd)
        ops.loss_grad(prd, tar, qw, (coef * gout).contiguous(), dprd)
        return dprd, None, None, None, None, None


class _FusedGeoL2(torch.autograd.Function):
    """LossHandler value from the quadrature sums the head's un-patchify epilogue produced (SWV2_EPI_UNPATCH_LOSS): one
    loss_finalize launch; backward hands d loss / d sums[..., 0] to the head's backward, which scales the stored residual."""

    @staticmethod
    def forward(ctx, sums, chw, absolute, squared):
        loss, coef = ops.loss_finalize_sums(sums, chw, absolute, squared)
        ctx.layers = sums.shape[0]
        ctx.save_for_backward(coef)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (coef,) = ctx.saved_tensors
        ds = torch.zeros((ctx.layers,) + tuple(coef.shape) + (2,), dtype=torch.float32, device=coef.device)
        ds[..., 0] = 0.5 * gout * coef               # coef = 2 d loss / d S0 (every layer of the sums alike); the relative
        return ds, None, None, None                  # losses' d / d S1 has no consumer (the target needs no gradient)


class _LossCtx:
    """one forward's hand-over between LossHandler.fused_with and the model head"""

    def __init__(self, tar, qw):
        self.tar, self.qw, self.y, self.sums = tar, qw, None, None

    def fits(self, x, net):
        t = self.tar
        return (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad and t.dim() == 4 and
                t.shape[0] == x.shape[0] and t.shape[1] == net.out_chans and tuple(t.shape[2:]) == tuple(x.shape[2:]) and
                self.qw.numel() == t.shape[2] and t.shape[3] % 4 == 0 and (t.shape[2] // 4) * (t.shape[3] // 4) >= 32 and
                t.numel() < 2 ** 32 and x.numel() < 2 ** 32)

    def offer(self, y, sums):
        self.y, self.sums = y, sums

    # rollouts (MultiStepWrapper): every step's head offers the sums of its Cout channels of the concatenated prediction
    def fits_step(self, x, net, result, coff):
        t = self.tar
        return (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad and t.dim() == 4 and
                t.shape[0] == x.shape[0] and t.shape[1] == result.shape[1] and coff + net.out_chans <= t.shape[1] and
                tuple(t.shape[2:]) == tuple(x.shape[2:]) and self.qw.numel() == t.shape[2] and t.shape[3] % 4 == 0 and
                (t.shape[2] // 4) * (t.shape[3] // 4) >= 32 and t.numel() < 2 ** 32 - 1024 and x.numel() < 2 ** 32 - 1024 and
                # what the fused rollout path itself needs (else the two-pass loss, as before -- ADVICE r5): the dump offset is a C int,
                # swv2_loss_resid_to_image stages 16 rows of Cout * 16 (+ 8) bf16 in 64 KB of LDS, and the concatenated prediction must
                # carry the epilogue's scratch behind it
                result.numel() < 2 ** 31 - 1024 and 16 * (net.out_chans * 16 + 8) * 2 <= 65536 and self._slack_behind(result) >= ops.L.LOSS_DUMP_BYTES)

    @staticmethod
    def _slack_behind(t):
        """allocated bytes behind the last element of a (view of a) tensor"""
        base = t._base if t._base is not None else t
        return base.data_ptr() + base.numel() * base.element_size() - (t.data_ptr() + t.numel() * t.element_size())

    def offer_step(self, result, coff, sums):
        if self.steps is None or self.result_ptr != result.data_ptr():
            self.steps, self.result_ptr, self.result_shape = [], result.data_ptr(), tuple(result.shape)
        self.steps.append((coff, sums))

    steps, result_ptr, result_shape = None, None, None

    def rollout_sums(self, prd, n_out):
        """[SLICES, B, S * Cout, 2] if every channel block of `prd` (the rollout's concatenated prediction) was offered, else None"""
        if not self.steps or prd.data_ptr() != self.result_ptr or tuple(prd.shape) != self.result_shape:
            return None
        steps = sorted(self.steps, key=lambda cs: cs[0])
        if [c for c, _ in steps] != list(range(0, prd.shape[1], n_out)) or any(s.shape[2] != n_out for _, s in steps):
            return None
        return torch.cat([s for _, s in steps], dim=2)


def _core_net(model):
    """the SwinTransformerV2Cr under DDP (`.module`) and the step wrappers (`.model`)"""
    m, seen = model, 0
    while seen < 4 and not hasattr(m, "_features_bhwc"):
        m = getattr(m, "module", None) or getattr(m, "model", None)
        if m is None:
            return None
        seen += 1
    return m if hasattr(m, "_features_bhwc") else None


def auto_channel_weights(channel_names, n_out):
    """losses.py:57-68"""
    w = torch.ones(n_out, dtype=torch.float32)
    for c, chn in enumerate(channel_names):
        if chn in ['u10m', 'v10m', 'u100m', 'v100m', 'tp', 'sp', 'msl', 'tcwv']:
            w[c] = 0.1
        elif chn in ['t2m', '2d']:
            w[c] = 1.0
        elif chn[0] in ['z', 'u', 'v', 't', 'r', 'q']:
            w[c] = 0.001 * float(chn[1:])
        else:
            w[c] = 0.01
    return w


def load_stats(params):
    """(global_stds, time_diff_stds) as [1, N, 1, 1] float32 arrays (losses.py:91-92 loads them from params.*_path).
    Without the ERA5 statistics files (synthetic-data runs: no dataset in the image) the stand-ins of SURVEY.md 8(d) are
    used: unit global stds and a fixed, seeded positive vector of time-difference stds."""
    import os
    gp, tp = str(getattr(params, 'global_stds_path', '')), str(getattr(params, 'time_diff_stds_path', ''))
    if os.path.isfile(gp) and os.path.isfile(tp):
        return np.load(gp).astype(np.float32), np.load(tp).astype(np.float32)
    n = max(len(getattr(params, 'channel_names', [])), int(np.max(np.asarray(params.out_channels))) + 1)
    rng = np.random.RandomState(333)
    return np.ones((1, n, 1, 1), np.float32), (0.05 + 0.45 * rng.rand(1, n, 1, 1)).astype(np.float32)


class LossHandler(nn.Module):
    """Wrapper class that handles computing losses: `LossHandler(params)(prd, tar, inp)` (losses.py:30-150)."""

    def __init__(self, params):
        super().__init__()
        self.n_future = params.n_future
        self.img_shape = (params.img_shape_x, params.img_shape_y)
        self.loss_type = params.loss
        flags = set(params.loss.split())
        if 'pole-masked' in flags:
            raise ValueError("pole-masked losses hit an undefined name in the reference (grids.py:97-99)")
        if 'l2' not in flags:
            raise ValueError(f"Unknown / unsupported loss function: {self.loss_type} (l2 family only)")
        if 'weighted' in flags:
            if params.channel_weights == 'auto':
                cw = auto_channel_weights(params.channel_names, params.n_out_channels)
            else:
                cw = torch.Tensor(params.channel_weights).float()
        else:
            cw = torch.ones(params.n_out_channels, dtype=torch.float32)
        cw = cw.reshape(1, -1, 1, 1)
        cw = cw / torch.sum(cw)
        self.absolute = 'absolute' in flags
        # plain 'l2' never forwards `squared` (losses.py:112-113), and is sphere-weighted all the same (quirk 9)
        self.squared = ('squared' in flags) and ('geometric' in flags)
        if 'temp-std' in flags:
            eps = 1e-6
            oc = np.asarray(params.out_channels)
            gs_np, td_np = load_stats(params)
            gstd = torch.from_numpy(gs_np).reshape(1, -1, 1, 1)[:, oc]
            tstd = np.sqrt(params.dt) * torch.from_numpy(td_np).reshape(1, -1, 1, 1)[:, oc]
            tvw = gstd / (tstd + eps)
            if 'squared' in flags:
                tvw = tvw ** 2
            cw = cw * tvw
        self.register_buffer('channel_weights', cw.float())
        if getattr(params, 'model_grid_type', 'equiangular') == 'legendre_gauss':
            raise ValueError("legendre-gauss quadrature needs torch_harmonics (not used by config/swin.yaml)")
        self.register_buffer('quad_rows', naive_quadrature_weights(self.img_shape[0], self.img_shape[1], True).contiguous())
        ms = torch.ones(self.n_future + 1, dtype=torch.float32) / float(self.n_future + 1)
        self.register_buffer('multistep_weight', ms.reshape(-1, 1, 1, 1))

    _fused = None

    def fused_with(self, model, tar: torch.Tensor):
        """`with loss_obj.fused_with(model, tar): gen = model(inp)` then `loss_obj(gen, tar, inp)` as usual: the model's head
        evaluates this loss's quadrature sums in the epilogue that writes `gen` and keeps the weighted residual for its own
        backward, so the prediction is not read back by the loss (2 passes) nor a gradient of its size written and re-read
        (losses.py:188-206 on the output of swinv2_global.py:784-802).  Purely an accelerator: same value, same gradients;
        when the shapes do not fit (rollouts, skip inputs that need gradients, eval) the two-pass kernels run instead."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            net = _core_net(model) if (self.training and os.environ.get("SWV2_LOSS_IN_HEAD", "1") != "0") else None
            # (rollouts: every step's head takes its channel block of the target; SWV2_LOSS_IN_HEAD_ROLLOUT=0 keeps the two-pass kernels)
            if net is None or (self.n_future != 0 and os.environ.get("SWV2_LOSS_IN_HEAD_ROLLOUT", "1") == "0"):
                yield
                return
            lc = _LossCtx(tar, self.quad_rows)
            net._loss_ctx, self._fused = lc, lc
            try:
                yield
            finally:
                net._loss_ctx = None
        return cm()

    def forward(self, prd: torch.Tensor, tar: torch.Tensor, inp: torch.Tensor = None):
        lc, self._fused = self._fused, None
        rsums = None
        if lc is not None and lc.steps and lc.tar is tar and self.training and prd.dtype == torch.float32:
            rsums = lc.rollout_sums(prd, prd.shape[1] // (self.n_future + 1))
        if rsums is not None or (lc is not None and lc.y is not None and lc.tar is tar and self.training and
                                 (lc.y is prd or (prd.data_ptr() == lc.y.data_ptr() and prd.shape == lc.y.shape and prd.dtype == lc.y.dtype))):
            # (the product of two constant buffers: cached, one tiny launch per step less; re-made when a buffer moved or changed)
            key = (self.channel_weights.data_ptr(), self.channel_weights._version, self.multistep_weight.data_ptr(), self.multistep_weight._version)
            if getattr(self, "_chw_key", None) != key:
                self._chw_flat = (self.channel_weights * self.multistep_weight).reshape(-1).contiguous().float()
                self._chw_key = key
            return _FusedGeoL2.apply(lc.sums if rsums is None else rsums, self._chw_flat, self.absolute, self.squared)
        chw = self.channel_weights
        if self.training:
            chw = (chw * self.multistep_weight).reshape(1, -1)
        else:
            chw = chw.reshape(1, -1)
        if prd.is_cuda and prd.requires_grad and not tar.requires_grad and os.environ.get("SWV2_LOSS_FUSED", "1") != "0":
            return _GeoL2.apply(prd, tar, self.quad_rows, chw.reshape(-1).contiguous().float(), self.absolute, self.squared)
        sums = _QuadSums.apply(prd, tar, self.quad_rows)
        norms = sums[..., 0]
        if not self.absolute:
            norms = norms / sums[..., 1]
        if not self.squared:
            norms = torch.sqrt(norms)
        return torch.sum(chw * norms)
