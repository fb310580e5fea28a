"""`import models.resnet38d` (models/revise_net.py:6, models/net_cls.py:5): Net, ResBlock, ResBlock_bot, convert_mxnet_to_torch (revise_pseudo_labels.py:180)."""
import _pistoseg_compat  # noqa: F401
from pistoseg_amd.resnet38d import Net, ResBlock, ResBlock_bot, convert_mxnet_to_torch  # noqa: F401
