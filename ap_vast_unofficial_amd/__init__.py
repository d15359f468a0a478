"""MI355X-native AP-VAST filter engine (subband hot path) behind the reference's
``apvast`` class surface.  The compute path is libapvast_hip.so (hand-written HIP
for gfx950) reached through a C ABI; there is no CPU fallback."""
from . import _capi
from ._capi import Engine, ApvError

__all__ = ["Engine", "ApvError", "_capi"]
