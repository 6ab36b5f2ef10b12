"""SAM assembly for MFNet: `sam_model_registry[name](args, checkpoint=path) -> Sam`
(reference: medsam/models/sam/build_sam.py:16-152, medsam/models/sam/modeling/sam.py:18-49).

Only the image encoder is ever executed by MFNet. SAM's prompt encoder and mask decoder are built by
the reference, sit in every MFNet state_dict and stay `requires_grad=True` without ever receiving a
gradient, so here they are PARAMETER CONTAINERS with the identical key/shape
layout (so checkpoints round-trip key-for-key) and no forward.
"""
from __future__ import annotations

from functools import partial
from pathlib import Path
from typing import Dict, Sequence, Tuple

import torch
import torch.nn as nn

from .vit import ImageEncoderViT


class _ParamTree(nn.Module):
    """nn.Module hierarchy generated from {dotted.key: (shape, kind)}; kind 'p' parameter, 'b' buffer."""

    def __init__(self, spec: Dict[str, Tuple[Sequence[int], str]]):
        super().__init__()
        groups: Dict[str, dict] = {}
        for key, (shape, kind) in spec.items():
            head, _, rest = key.partition(".")
            if rest:
                groups.setdefault(head, {})[rest] = (shape, kind)
            else:
                t = torch.empty(*shape)
                if len(shape) >= 2:
                    nn.init.normal_(t, std=0.02)
                elif head == "weight":
                    nn.init.ones_(t)
                else:
                    nn.init.zeros_(t)
                if kind == "p":
                    self.register_parameter(head, nn.Parameter(t))
                else:
                    self.register_buffer(head, t)
        for head, sub in groups.items():
            self.add_module(head, _ParamTree(sub))

    def forward(self, *a, **k):
        raise NotImplementedError("SAM's prompt encoder / mask decoder are never called by MFNet "
                                  "(unetformer_mmsam.py:533-563); only their parameters are kept")


def _lin(spec, p, o, i):
    spec[p + ".weight"] = ((o, i), "p")
    spec[p + ".bias"] = ((o,), "p")


def _ln(spec, p, c):
    spec[p + ".weight"] = ((c,), "p")
    spec[p + ".bias"] = ((c,), "p")


def prompt_encoder_spec(embed_dim=256, mask_in_chans=16):
    s = {"pe_layer.positional_encoding_gaussian_matrix": ((2, embed_dim // 2), "b")}
    for i in range(4):
        s[f"point_embeddings.{i}.weight"] = ((1, embed_dim), "p")
    s["not_a_point_embed.weight"] = ((1, embed_dim), "p")
    c4 = mask_in_chans // 4
    s["mask_downscaling.0.weight"] = ((c4, 1, 2, 2), "p"); s["mask_downscaling.0.bias"] = ((c4,), "p")
    _ln(s, "mask_downscaling.1", c4)
    s["mask_downscaling.3.weight"] = ((mask_in_chans, c4, 2, 2), "p"); s["mask_downscaling.3.bias"] = ((mask_in_chans,), "p")
    _ln(s, "mask_downscaling.4", mask_in_chans)
    s["mask_downscaling.6.weight"] = ((embed_dim, mask_in_chans, 1, 1), "p"); s["mask_downscaling.6.bias"] = ((embed_dim,), "p")
    s["no_mask_embed.weight"] = ((1, embed_dim), "p")
    return s


def mask_decoder_spec(dim=256, num_multimask_outputs=1, depth=2, mlp_dim=2048, iou_hidden=256, iou_depth=3):
    s = {}

    def attn(p, down):
        inner = dim // down
        for n in ("q_proj", "k_proj", "v_proj"):
            _lin(s, f"{p}.{n}", inner, dim)
        _lin(s, f"{p}.out_proj", dim, inner)

    for i in range(depth):
        p = f"transformer.layers.{i}"
        attn(p + ".self_attn", 1)
        _ln(s, p + ".norm1", dim)
        attn(p + ".cross_attn_token_to_image", 2)
        _ln(s, p + ".norm2", dim)
        _lin(s, p + ".mlp.layers.0.0", mlp_dim, dim)
        _lin(s, p + ".mlp.fc", dim, mlp_dim)
        _ln(s, p + ".norm3", dim)
        _ln(s, p + ".norm4", dim)
        attn(p + ".cross_attn_image_to_token", 2)
    attn("transformer.final_attn_token_to_image", 2)
    _ln(s, "transformer.norm_final_attn", dim)
    ntok = max(4, num_multimask_outputs)          # mask_decoder.py:49
    s["iou_token.weight"] = ((1, dim), "p")
    s["mask_tokens.weight"] = ((ntok, dim), "p")
    s["output_upscaling.0.weight"] = ((dim, dim // 4, 2, 2), "p"); s["output_upscaling.0.bias"] = ((dim // 4,), "p")
    _ln(s, "output_upscaling.1", dim // 4)
    s["output_upscaling.3.weight"] = ((dim // 4, dim // 8, 2, 2), "p"); s["output_upscaling.3.bias"] = ((dim // 8,), "p")
    for i in range(ntok):
        _lin(s, f"output_hypernetworks_mlps.{i}.layers.0", dim, dim)
        _lin(s, f"output_hypernetworks_mlps.{i}.layers.1", dim, dim)
        _lin(s, f"output_hypernetworks_mlps.{i}.layers.2", dim // 8, dim)
    dims = [dim] + [iou_hidden] * (iou_depth - 1) + [ntok]
    for i in range(iou_depth):
        _lin(s, f"iou_prediction_head.layers.{i}", dims[i + 1], dims[i])
    return s


class Sam(nn.Module):
    """sam.py:18-49: container exposing .image_encoder / .prompt_encoder / .mask_decoder."""
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, args, image_encoder, prompt_encoder, mask_decoder,
                 pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)):
        super().__init__()
        self.args = args
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)

    @property
    def device(self):
        return self.pixel_mean.device


def _build_sam(args, encoder_embed_dim, encoder_depth, encoder_num_heads, encoder_global_attn_indexes, checkpoint=None):
    """build_sam.py:60-152. Checkpoint loading keeps only same-name-same-shape tensors (:145-151).
    The reference's interactive download branch (:113-142) is NOT reproduced: a missing checkpoint file
    raises FileNotFoundError (there is no network on the target machines)."""
    prompt_embed_dim = 256
    sam = Sam(
        args,
        image_encoder=ImageEncoderViT(
            args=args, depth=encoder_depth, embed_dim=encoder_embed_dim, img_size=args.image_size, mlp_ratio=4,
            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=encoder_num_heads, patch_size=16,
            qkv_bias=True, use_rel_pos=True, global_attn_indexes=encoder_global_attn_indexes, window_size=14,
            out_chans=prompt_embed_dim),
        prompt_encoder=_ParamTree(prompt_encoder_spec(prompt_embed_dim, 16)),
        mask_decoder=_ParamTree(mask_decoder_spec(prompt_embed_dim, args.multimask_output)),
    )
    sam.eval()
    if checkpoint is not None:
        checkpoint = Path(checkpoint)
        if not checkpoint.exists():
            raise FileNotFoundError(f"SAM checkpoint {checkpoint} not found (download it beforehand; the "
                                    "reference's interactive download is not reproduced)")
        with open(checkpoint, "rb") as f:
            state_dict = torch.load(f, map_location="cpu", weights_only=True)
        own = sam.state_dict()
        keep = {k: v for k, v in state_dict.items() if k in own and own[k].shape == v.shape}
        sam.load_state_dict(keep, strict=False)
    return sam


def build_sam_vit_h(args=None, checkpoint=None):
    return _build_sam(args, 1280, 32, 16, [7, 15, 23, 31], checkpoint)


def build_sam_vit_l(args, checkpoint=None):
    return _build_sam(args, 1024, 24, 16, [5, 11, 17, 23], checkpoint)


def build_sam_vit_b(args, checkpoint=None):
    return _build_sam(args, 768, 12, 12, [2, 5, 8, 11], checkpoint)


build_sam = build_sam_vit_h


def _build_sam_vit_tiny_test(args, checkpoint=None):
    """depth-2 / dim-128 encoder (one windowed + one global block) used by the golden-vector tests."""
    return _build_sam(args, 128, 2, 2, [1], checkpoint)


sam_model_registry = {
    "vit_tiny_test": _build_sam_vit_tiny_test,
    "default": build_sam_vit_b,
    "vit_h": build_sam_vit_h,
    "vit_l": build_sam_vit_l,
    "vit_b": build_sam_vit_b,
}
