"""`models` package of the reference tree, resolved to the MI355X mirrors (see ../_pistoseg_compat.py)."""
import _pistoseg_compat  # noqa: F401
