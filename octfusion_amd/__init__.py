"""octfusion_amd: MI355X-native (gfx950) implementation of OctFusion's diffusion
U-Net denoising hot path.

Host side = thin Python mirrors of the reference's nn.Module interface
(models/networks/modules.py, dual_octree.py, graph_unet_{hr,lr,union}.py); all
compute on the dual-octree path goes through the C ABI in include/ofx.h
(libofx.so, hand-written HIP).  There is no CPU / eager fallback.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
