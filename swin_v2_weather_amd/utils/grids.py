"""Quadrature weights on the lat-lon grid (reference utils/grids.py:62-117, 'naive' rule only: the other rules need
torch_harmonics and are selected by no config)."""
import math

import torch


def naive_quadrature_weights(H: int, W: int, normalize: bool = True) -> torch.Tensor:
    """[H] latitude weights: sin(linspace(0, pi, H)) * dA rescaled to sum 4 pi over the [H, W] grid, / 4 pi if
    `normalize` (grids.py:68-76, 93-94).  Every longitude has the same weight, so one value per row is stored."""
    jac = torch.clamp(torch.sin(torch.linspace(0, math.pi, H)), min=0.0)
    dA = (2 * math.pi / W) * (math.pi / H)
    q = dA * jac
    q = q * (4.0 * math.pi) / (q.sum() * W)
    if normalize:
        q = q / (4.0 * math.pi)
    return q


class GridQuadrature(torch.nn.Module):
    """Integrates over the last two axes with the latitude weights (grids.py:62-117).  Host-side helper; the training
    loss uses the fused kernels in losses.py."""

    def __init__(self, quadrature_rule, img_shape, crop_shape=None, crop_offset=(0, 0), normalize=False, pole_mask=None):
        super().__init__()
        if quadrature_rule != 'naive':
            raise ValueError(f"Unknown quadrature rule {quadrature_rule} (only 'naive' is available without torch_harmonics)")
        q = naive_quadrature_weights(img_shape[0], img_shape[1], normalize).unsqueeze(1).repeat(1, img_shape[1])
        if pole_mask:
            q[:pole_mask, :] = 0.0
            q[img_shape[0] - pole_mask:, :] = 0.0
        if crop_shape is not None:
            q = q[crop_offset[0]:crop_offset[0] + crop_shape[0], crop_offset[1]:crop_offset[1] + crop_shape[1]]
        self.register_buffer('quad_weight', q.contiguous().reshape(1, 1, *q.shape))

    def forward(self, x):
        return torch.sum(x * self.quad_weight, dim=(-2, -1))
