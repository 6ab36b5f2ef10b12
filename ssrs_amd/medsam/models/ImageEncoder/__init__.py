"""medsam.models.ImageEncoder drop-in (the two dual-modality blocks)."""
from ....vit import LoraBlock  # noqa: F401
from ....vit_adapter import AdapterFusionBlock  # noqa: F401
