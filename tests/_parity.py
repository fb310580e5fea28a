"""Shared parity bookkeeping: mask indices are asserted bit-exact EXCEPT at pixels whose top-2 oracle scores are closer than the
measured f32 score error (either index is then a correct argmax of scores that agree within the tolerance).  The number of such
tie-excused pixels is printed and capped, so that a regression from a handful to thousands cannot pass unnoticed."""
import math

TIE_CAP_FRACTION = 1e-4   # of the pixels compared
TIE_CAP_FLOOR = 2         # small maps: a couple of near-tie pixels may flip


def tie_cap(pixels: int) -> int:
    return max(TIE_CAP_FLOOR, math.ceil(TIE_CAP_FRACTION * pixels))


def assert_tie_excused(what: str, n_diff: int, pixels: int, all_within_gap: bool) -> None:
    print(f"[parity] {what}: {n_diff} of {pixels} mask pixels differ from the oracle (tie-excused cap {tie_cap(pixels)})")
    assert all_within_gap, f"{what}: {n_diff} mask pixels differ beyond the tie tolerance"
    assert n_diff <= tie_cap(pixels), f"{what}: {n_diff} tie-excused pixels exceed the cap of {tie_cap(pixels)} (of {pixels})"
