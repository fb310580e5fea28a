"""pistoseg_amd -- MI355X-native (gfx950) implementation of the PistoSeg segmentation hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the
path runs in hand-written HIP kernels behind the C-ABI of libpistoseg_hip.so (include/pistoseg_hip.h).
"""
__version__ = "0.1.0"
