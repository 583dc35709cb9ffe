"""Logging set-up for the trainer.  The three entry points keep the names the reference's train.py calls
(utils/logging_utils.py: config_logger, log_to_file, log_versions); what they do is this package's own:
one record layout for console and file, a file handler that is attached once per path (a resumed run in the same process
does not duplicate every line), and a version banner that names what the numbers depend on here -- the HIP library's ABI
revision and source hash, the ROCm runtime torch was built against, the device."""
import logging
import pathlib

RECORD = "%(asctime)s - %(name)s - %(levelname)s - %(message)s"


def config_logger(log_level=logging.INFO):
    """Console logging with the common record layout (no-op for the root logger's handlers if it is configured already)."""
    logging.basicConfig(level=log_level, format=RECORD)


def log_to_file(logger_name=None, log_level=logging.INFO, log_filename="tensorflow.log"):
    """Mirror `logger_name` (None: the root logger) into `log_filename`; its directory is created on demand."""
    path = pathlib.Path(log_filename).resolve()
    path.parent.mkdir(parents=True, exist_ok=True)
    target = logging.getLogger(logger_name)
    for h in target.handlers:
        if isinstance(h, logging.FileHandler) and pathlib.Path(h.baseFilename) == path:
            h.setLevel(log_level)
            return
    handler = logging.FileHandler(path)
    handler.setLevel(log_level)
    handler.setFormatter(logging.Formatter(RECORD))
    target.addHandler(handler)


def log_versions():
    """What a logged throughput or loss depends on: torch / ROCm, the HIP library (ABI revision, source hash), the device."""
    import torch
    from .. import _lib
    lines = [f"torch {torch.__version__} (hip {getattr(torch.version, 'hip', None)})",
             f"libswv2 ABI {_lib.ABI_VERSION}, sources {_lib.source_hash()}"]
    if torch.cuda.is_available():
        lines.append(f"device {torch.cuda.get_device_name(0)} x {torch.cuda.device_count()}")
    bar = "-" * 16
    logging.info("%s versions %s", bar, bar)
    for ln in lines:
        logging.info(ln)
    logging.info("-" * 42)
