def get_data_loader_distributed(params, location, distributed, train):
    """Same entry point as the reference (utils/__init__.py:1-6).  DALI is NVIDIA-only and h5py / ERA5 files are not
    present here: every `data_loader_config` maps to the synthetic-field loader of `data_loader_era5` (which keeps the
    reference's dataset index arithmetic and tensor contract)."""
    from .data_loader_era5 import get_data_loader
    return get_data_loader(params, location, distributed, train)
