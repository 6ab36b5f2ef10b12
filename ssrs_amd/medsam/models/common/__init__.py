"""medsam.models.common drop-in (Adapter, LayerNorm2d, loralib)."""
from ....vit import Adapter, LayerNorm2d  # noqa: F401
from . import loralib  # noqa: F401
