"""Drop-in for the reference's ``graphs/stylegan_v2_real/op`` package (op/__init__.py:1-2): same three names, backed
by libl2i_hip.so instead of the JIT-built CUDA extensions."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d

__all__ = ['FusedLeakyReLU', 'fused_leaky_relu', 'upfirdn2d']
