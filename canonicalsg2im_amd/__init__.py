"""MI355X-native training hot path of CanonicalSg2Im (scene graph -> layout -> image).

Host side is Python on PyTorch-ROCm; every inner kernel is hand-written HIP for
gfx950 behind the C ABI declared in `include/csg_hip.h` (`libcsg_hip.so`).
There is no CPU fallback: importing `canonicalsg2im_amd.ops` without the built
library raises.
"""
__version__ = "0.1.0"
