"""evreal_amd -- MI355X-native hot path of EVREAL (events->voxel grid, recurrent-UNet
inference, per-frame metrics) behind the reference's eval.py plugin surface.

The compute lives in csrc/ (hand-written HIP for gfx950, exported through the C ABI declared
in include/evreal_hip.h).  There is no CPU fallback: every op raises if libevreal_hip.so or a
GPU is missing.
"""
__version__ = "0.1.0"
