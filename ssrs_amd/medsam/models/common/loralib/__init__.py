"""loralib drop-in: the two layers MFNet uses (reference: medsam/models/common/loralib/layers.py)."""
from .....vit import Linear, LoRALayer, MergedLinear  # noqa: F401
