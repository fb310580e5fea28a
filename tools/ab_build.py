"""A/B builds of the debug library with compile-time defines:  python tools/ab_build.py <tag> NAME=VALUE ...
-> pistoseg_amd/libpistoseg_hip_debug_<tag>.so; run a tool against it with PISTOSEG_HIP_DEBUG_LIB=pistoseg_amd/libpistoseg_hip_debug_<tag>.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pistoseg_amd import build
print(build.build_variant(sys.argv[1], sys.argv[2:]))
