"""PreProcessor with the reference's surface (utils/preprocess_utils.py:5-69): moves a loader batch to the device as
fp32, appends the input cos-zenith channel and the static features (one-hot land mask: 2 channels, then standardised
orography: 1 channel) and returns (inp, tar, target_zenith | None).  The invariant files (orog.nc / lsm.h5) are not
available here, so the static fields are seeded synthetic stand-ins: land mask ~ Bernoulli(0.3), orography ~ N(0,1)."""
import torch
import torch.nn as nn


class PreProcessor(nn.Module):
    def __init__(self, params, device):
        super().__init__()
        self.params, self.device = params, device
        imgx, imgy = params.img_size
        g = torch.Generator().manual_seed(4242)
        static_features = None
        if self.params.add_landmask:
            lsm = (torch.rand(721, 1440, generator=g) < 0.3).long()
            lsm = torch.permute(torch.nn.functional.one_hot(lsm, 2), (2, 0, 1)).to(torch.float32)
            static_features = lsm.reshape(1, 2, 721, 1440)[:, :, :imgx, :imgy]
        if self.params.add_orography:
            oro = torch.randn(721, 1440, generator=g).reshape(1, 1, 721, 1440)[:, :, :imgx, :imgy]
            oro = (oro - torch.mean(oro)) / (torch.std(oro) + 1.0e-6)
            static_features = oro if static_features is None else torch.cat([static_features, oro], dim=1)
        self.do_add_static_features = static_features is not None
        if self.do_add_static_features:
            self.register_buffer("static_features", static_features.contiguous(), persistent=False)

    def forward(self, data):
        if self.params.add_zenith:
            inp, tar, izen, tzen = map(lambda x: x.to(self.device, dtype=torch.float), data)
            inp = torch.cat([inp, izen], dim=1)
        else:
            inp, tar = map(lambda x: x.to(self.device, dtype=torch.float), data)
            tzen = None
        if self.do_add_static_features:
            inp = torch.cat([inp, self.static_features.expand(inp.shape[0], -1, -1, -1)], dim=1)
        return inp, tar, tzen
