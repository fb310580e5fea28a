"""`from models.revise_net import Net` (revise_pseudo_labels.py:28, infer_revise_masks.py:20)."""
import _pistoseg_compat  # noqa: F401
from pistoseg_amd.revise_net import Net  # noqa: F401
