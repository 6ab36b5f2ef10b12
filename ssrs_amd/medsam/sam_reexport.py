from ..sam import (Sam, build_sam, build_sam_vit_b, build_sam_vit_h, build_sam_vit_l,  # noqa: F401
                   sam_model_registry)
