from .helpers import get_model, SingleStepWrapper, MultiStepWrapper  # noqa: F401
from .swinv2_global import swinv2net, swin_from_yaml, SwinTransformerV2Cr  # noqa: F401
