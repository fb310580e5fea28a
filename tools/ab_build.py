"""A/B builds of the debug library with compile-time defines:  python tools/ab_build.py <tag> NAME=VALUE ...
-> pistoseg_amd/libpistoseg_hip_debug_<tag>.so; run a tool against it with PISTOSEG_HIP_DEBUG_LIB=pistoseg_amd/libpistoseg_hip_debug_<tag>.so.
With --product as the first argument: the product library's flags -> pistoseg_amd/libpistoseg_hip_<tag>.so, for whole-step A/Bs:
PISTOSEG_HIP_LIB=pistoseg_amd/libpistoseg_hip_<tag>.so python bench.py ...  (a sustained single-kernel loop on random data sits at the board's power cap,
NOTES 7.28, and hides cycle-level gains that the real step, ~10 % under the cap, still shows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pistoseg_amd import build
argv = sys.argv[1:]
product = argv[:1] == ["--product"]
if product:
    argv = argv[1:]
print(build.build_variant(argv[0], argv[1:], product=product))
