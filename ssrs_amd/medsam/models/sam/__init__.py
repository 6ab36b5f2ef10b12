"""medsam.models.sam drop-in (reference: medsam/models/sam/__init__.py, build_sam.py:52-57)."""
from ...sam_reexport import *  # noqa: F401,F403
