"""maest_amd: MI355X-native (gfx950) implementation of the MAEST mel -> patchout-ViT hot path,
behind the reference's Python surface (``from maest import get_maest`` -> ``from maest_amd import get_maest``)."""
from .maest import MAEST, get_maest  # noqa: F401

__all__ = ["get_maest", "MAEST"]
