"""`from models.mosaic_module import MosaicModule` (mosaic_train.py:14, infer_pseudo_masks.py:17, segmentation_test.py:14)."""
import _pistoseg_compat  # noqa: F401
from pistoseg_amd.segmentation_module import MosaicModule  # noqa: F401
