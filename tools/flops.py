#!/usr/bin/env python3
"""Algorithmic FLOPs of the MFNet encoder: real tokens only, multiply-add = 2 FLOPs.
    python tools/flops.py            # table for ViT-B/L/H, 256^2 and 1024^2, MMLoRA / MMAdapter"""
CFG = {"vit_b": (768, 12, 12, 4), "vit_l": (1024, 24, 16, 4), "vit_h": (1280, 32, 16, 4)}   # dim, depth, heads, #global blocks


def encoder_fwd_gf(name, size=256, mod="sam_lora", r=4):
    D, depth, heads, nglob = CFG[name]
    hd = D // heads
    S = size // 16
    T = S * S
    gemm = 2 * T * D * 3 * D + 2 * T * D * D + 2 * 2 * T * D * 4 * D
    attn_g = 4 * heads * T * T * hd + 4 * heads * T * S * hd
    attn_w = 4 * heads * T * 196 * hd + 4 * heads * T * 14 * hd
    if mod == "sam_lora":    # qkv: A (2r x D) + B on q,v ; lin1 / lin2: two modalities' low-rank paths each
        extra = 2 * T * D * 2 * r + 2 * T * 2 * D * r + 2 * (2 * T * D * r + 2 * T * 4 * D * r) * 2
    else:                    # 2 adapters per modality and block: D -> D/4 -> D
        extra = 2 * (2 * T * D * (D // 4) * 2)
    per_mod = depth * (gemm + extra) + nglob * attn_g + (depth - nglob) * attn_w
    per_mod += 2 * T * 768 * D                                   # patch embed
    per_mod += 2 * T * D * 256 + 2 * T * 256 * 256 * 9           # neck
    return 2 * per_mod / 1e9


def encoder_train_gf(name, size=256, mod="sam_lora"):
    """fwd + dgrad of every GEMM (frozen weights: no wgrad; patch embed: no backward) + attention backward
    (~2.5x its forward) + PEFT weight gradients (= one more pass over the PEFT products)."""
    D, depth, heads, nglob = CFG[name]
    S = size // 16
    T = S * S
    hd = D // heads
    f = encoder_fwd_gf(name, size, mod) * 1e9 / 2
    pe = 2 * T * 768 * D
    attn = nglob * (4 * heads * T * T * hd) + (depth - nglob) * (4 * heads * T * 196 * hd)
    dense = f - pe - attn
    return 2 * (pe + 2 * dense + 3.5 * attn) / 1e9


if __name__ == "__main__":
    for size in (256, 1024):
        for n in CFG:
            print(f"{n} {size}^2: fwd MMLoRA {encoder_fwd_gf(n, size):8.1f} GF/tile  MMAdapter {encoder_fwd_gf(n, size, 'sam_adpt'):8.1f}"
                  f"   train MMLoRA {encoder_train_gf(n, size):8.1f}  MMAdapter {encoder_train_gf(n, size, 'sam_adpt'):8.1f}")
