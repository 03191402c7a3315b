"""feabas_amd -- MI355X (gfx950) implementation of FEABAS's two hot paths.

Host-side modules keep the reference's call surface (``matcher.xcorr_fft``,
``common.masked_dog_filter``, ``optimizer.SLM`` / ``solve``, ``mesh.Mesh``) and
run the arithmetic through the C ABI of ``libfeabas_hip.so``
(``include/feabas_hip.h``).  No CPU fallback exists.
"""
from . import _lib                                  # noqa: F401
from . import constant                              # noqa: F401
from . import common, material, matcher, mesh, optimizer      # noqa: F401

__all__ = ['common', 'material', 'matcher', 'mesh', 'optimizer', 'constant']
