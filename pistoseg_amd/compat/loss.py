"""Drop-in for the reference's `loss.py` (`from loss import mIoUMask`: revise_pseudo_labels.py:4, segmentation_test.py:13,
models/segmentation_module.py:27): `mIoUMask` is the device-resident mirror (pistoseg_amd/metrics.py; reference loss.py:8-67);
anything else a `loss.py` further down sys.path defines stays visible."""
import _pistoseg_compat

_pistoseg_compat.overlay_next_on_path("loss", globals())
from pistoseg_amd.metrics import mIoUMask  # noqa: E402,F401
