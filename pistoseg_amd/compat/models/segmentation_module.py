"""`from models.segmentation_module import SegmentationModule` (segmentation_train.py:8, segmentation_test.py:29)."""
import _pistoseg_compat  # noqa: F401
from pistoseg_amd.segmentation_module import SegmentationModule  # noqa: F401
