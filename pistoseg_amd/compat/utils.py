"""Drop-in for the reference's `utils.py` (`import utils`: revise_pseudo_labels.py:25, infer_revise_masks.py:17): `PolyOptimizer`
(utils.py:166-187) becomes the fused-kernel mirror in pistoseg_amd/optim.py; the host-side helpers of the reference's own
`utils.py` (label parsing, `get_background`, `visualize`: CPU data plumbing, out of scope) stay visible when that file is
further down sys.path."""
import _pistoseg_compat

_pistoseg_compat.overlay_next_on_path("utils", globals())
from pistoseg_amd.optim import PolyOptimizer  # noqa: E402,F401
