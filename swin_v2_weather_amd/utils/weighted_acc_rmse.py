"""Validation metric used by the trainer (reference utils/weighted_acc_rmse.py:50-86): cos-latitude weighted RMSE.
Validation only -- not on the timed training path -- so this is plain torch."""
import torch


def weighted_rmse_torch(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """[n, c, h, w] x2 -> [c]: mean over the batch of sqrt(mean_hw(w_lat (pred - target)^2)), with
    w_lat = num_lat cos(lat_j) / sum_j cos(lat_j), lat_j = 90 - 180 j / (num_lat - 1) degrees (pi ~ 3.1416 as there)."""
    num_lat = pred.shape[2]
    j = torch.arange(0, num_lat, device=pred.device)
    coslat = torch.cos(3.1416 / 180.0 * (90.0 - j * 180.0 / float(num_lat - 1)))
    weight = (num_lat * coslat / coslat.sum()).reshape(1, 1, -1, 1)
    return torch.sqrt(torch.mean(weight * (pred - target) ** 2.0, dim=(-1, -2))).mean(dim=0)
