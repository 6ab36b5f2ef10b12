"""CPU oracle for the MFNet hot path (TEST INFRASTRUCTURE - never imported by the product).

A functional, plain-PyTorch fp32 restatement of the reference algorithm: every function takes a
flat ``state_dict``-style mapping ``sd`` plus a key prefix ``p`` and tensors, and returns tensors.
Backward passes come from torch autograd over this restatement. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Pinning: the reference has no tests / golden vectors of its own, so this
restatement is pinned against outputs of the reference itself, generated in the build container by
``tools/make_goldens.py`` (which imports /root/reference) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` replays them.

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------
def vit_cfg(name: str = "vit_b", mod: str = "sam_lora", mid_dim: Optional[int] = None,
            image_size: int = 1024, num_classes: int = 6, **over) -> dict:
    """Hyper-parameters per medsam/models/sam/build_sam.py:16-49,60-89 and medsam/cfg.py:11,22,47."""
    table = {
        "vit_b": dict(dim=768, depth=12, heads=12, global_idx=(2, 5, 8, 11)),
        "vit_l": dict(dim=1024, depth=24, heads=16, global_idx=(5, 11, 17, 23)),
        "vit_h": dict(dim=1280, depth=32, heads=16, global_idx=(7, 15, 23, 31)),
    }
    cfg = dict(table[name]) if name in table else {}
    cfg.update(mod=mod, mid_dim=mid_dim, image_size=image_size, window=14, patch=16,
               out_chans=256, num_classes=num_classes, decode_channels=64, dec_window=8,
               dec_heads=8, dropout=0.1)
    cfg.update(over)
    return cfg


# --------------------------------------------------------------------------------------------
# ViT pieces
# --------------------------------------------------------------------------------------------
def rel_pos_table(q_size: int, k_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """lora_block.py:248-278 (get_rel_pos): gather (q,k,hd) from the (L,hd) table, linearly
    interpolating the table when its length differs from 2*max(q,k)-1."""
    need = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != need:
        t = F.interpolate(rel_pos.t()[None], size=need, mode="linear")[0].t()
    else:
        t = rel_pos
    qc = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    kc = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    idx = (qc - kc) + (k_size - 1) * max(q_size / k_size, 1.0)
    return t[idx.long()]


def attention_core(q, k, v, rel_h, rel_w, hw: Tuple[int, int], scale: float):
    """lora_block.py:181-193,281-317: scores from SCALED q, decomposed rel-pos from UNSCALED q.
    q,k,v: (Bn, N, hd)."""
    H, W = hw
    attn = (q * scale) @ k.transpose(-2, -1)
    if rel_h is not None:
        Rh = rel_pos_table(H, H, rel_h)
        Rw = rel_pos_table(W, W, rel_w)
        Bn, _, hd = q.shape
        rq = q.reshape(Bn, H, W, hd)
        bh = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
        bw = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
        attn = (attn.view(Bn, H, W, H, W) + bh[..., :, None] + bw[..., None, :]).view(Bn, H * W, H * W)
    attn = attn.softmax(dim=-1)
    return attn @ v


def split_heads(qkv: torch.Tensor, heads: int):
    """lora_block.py:175-179: (B,N,3C) -> three (B*heads, N, hd)."""
    B, N, C3 = qkv.shape
    t = qkv.reshape(B, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    return t.reshape(3, B * heads, N, -1).unbind(0)


def merge_heads(o: torch.Tensor, B: int, heads: int, H: int, W: int):
    """lora_block.py:189: (B*heads,N,hd) -> (B,H,W,C)."""
    return o.view(B, heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)


def window_partition(x: torch.Tensor, ws: int):
    """lora_block.py:199-220: zero-pad bottom/right to a multiple of ws, cut into windows."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(w: torch.Tensor, ws: int, pad_hw, hw):
    """lora_block.py:223-245."""
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def merged_linear_qv(sd: SD, p: str, x: torch.Tensor, mod: str) -> torch.Tensor:
    """loralib/layers.py:244-268 (MergedLinear, enable_lora=[True,False,True]): shared dense
    qkv plus a per-modality rank-r update on the q and v thirds, scaled by 1/r. ``mod`` is 'x'|'y'."""
    W, b = sd[p + "weight"], sd[p + "bias"]
    A, Bm = sd[p + f"lora_A{mod}"], sd[p + f"lora_B{mod}"]
    r = A.shape[0] // 2
    C = W.shape[0] // 3
    out = F.linear(x, W, b)
    a = F.linear(x, A)                      # (..., 2r): first r feed q, last r feed v (conv1d groups=2)
    dq = a[..., :r] @ Bm[:C].t()
    dv = a[..., r:] @ Bm[C:].t()
    upd = torch.cat([dq, torch.zeros_like(dq), dv], dim=-1) * (1.0 / r)
    return out + upd


def lora_linear_mix(sd: SD, p: str, x: torch.Tensor, y: torch.Tensor):
    """loralib/layers.py:149-165 (Linear): shared dense layer plus cross-modal mix of the two
    low-rank updates weighted by the scalars lora_wx / lora_wy (no 1/r scaling)."""
    W, b = sd[p + "weight"], sd[p + "bias"]
    lx = x @ sd[p + "lora_Ax"].t() @ sd[p + "lora_Bx"].t()
    ly = y @ sd[p + "lora_Ay"].t() @ sd[p + "lora_By"].t()
    wx, wy = sd[p + "lora_wx"], sd[p + "lora_wy"]
    rx = F.linear(x, W, b) + wx * lx + (1 - wx) * ly
    ry = F.linear(y, W, b) + wy * ly + (1 - wy) * lx
    return rx, ry


def _ln(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-6):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def lora_block(sd: SD, p: str, x: torch.Tensor, y: torch.Tensor, heads: int, window: int):
    """lora_block.py:64-87 (LoraBlock.forward), x,y: (B,H,W,C)."""
    hd = x.shape[-1] // heads
    scale = hd ** -0.5
    outs = []
    for t, m in ((x, "x"), (y, "y")):
        s = t
        t = _ln(sd, p + "norm1.", t)
        H, W = t.shape[1:3]
        if window > 0:
            t, pad_hw = window_partition(t, window)
        Bw, h, w, C = t.shape
        qkv = merged_linear_qv(sd, p + "attn.qkv.", t.reshape(Bw, h * w, C), m)
        q, k, v = split_heads(qkv, heads)
        o = attention_core(q, k, v, sd.get(p + "attn.rel_h"), sd.get(p + "attn.rel_w"), (h, w), scale)
        o = merge_heads(o, Bw, heads, h, w)
        o = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        if window > 0:
            o = window_unpartition(o, window, pad_hw, (H, W))
        outs.append(s + o)
    x, y = outs
    hx, hy = lora_linear_mix(sd, p + "mlp.lin1.", _ln(sd, p + "norm2.", x), _ln(sd, p + "norm2.", y))
    mx, my = lora_linear_mix(sd, p + "mlp.lin2.", F.gelu(hx), F.gelu(hy))
    return x + mx, y + my


def adapter(sd: SD, p: str, x: torch.Tensor, skip: bool):
    """common/adapter.py:16-25: D -> D/4 -> GELU -> D bottleneck, optional skip."""
    h = F.linear(F.gelu(F.linear(x, sd[p + "D_fc1.weight"], sd[p + "D_fc1.bias"])),
                 sd[p + "D_fc2.weight"], sd[p + "D_fc2.bias"])
    return x + h if skip else h


def adapter_fusion_block(sd: SD, p: str, x: torch.Tensor, y: torch.Tensor, heads: int, window: int):
    """adapter_fusionblock.py:76-114 (AdapterFusionBlock.forward)."""
    hd = x.shape[-1] // heads
    scale = hd ** -0.5
    outs = []
    for t, ad in ((x, "Img_Adapter."), (y, "DSM_Adapter.")):
        s = t
        t = _ln(sd, p + "norm1.", t)
        H, W = t.shape[1:3]
        if window > 0:
            t, pad_hw = window_partition(t, window)
        Bw, h, w, C = t.shape
        qkv = F.linear(t.reshape(Bw, h * w, C), sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        q, k, v = split_heads(qkv, heads)
        o = attention_core(q, k, v, sd.get(p + "attn.rel_h"), sd.get(p + "attn.rel_w"), (h, w), scale)
        o = merge_heads(o, Bw, heads, h, w)
        o = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        o = adapter(sd, p + ad, o, skip=True)          # applied in window layout (:91-92)
        if window > 0:
            o = window_unpartition(o, window, pad_hw, (H, W))
        outs.append(s + o)
    x, y = outs
    xn, yn = _ln(sd, p + "norm2.", x), _ln(sd, p + "norm2.", y)
    ax = adapter(sd, p + "MLPx_Adapter.", xn, skip=False)
    ay = adapter(sd, p + "MLPy_Adapter.", yn, skip=False)

    def mlp(t):
        return F.linear(F.gelu(F.linear(t, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])),
                        sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
    wx, wy = sd[p + "wx_Adapter"], sd[p + "wy_Adapter"]
    x = x + mlp(xn) + wx * ax + (1 - wx) * ay
    y = y + mlp(yn) + wy * ay + (1 - wy) * ax
    return x, y


def layer_norm_2d(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-6):
    """common/layer_norm.py:12-17: LayerNorm over the channel dim of NCHW."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + "weight"][:, None, None] * x + sd[p + "bias"][:, None, None]


def image_encoder(sd: SD, p: str, x: torch.Tensor, y: torch.Tensor, cfg: dict):
    """sam/modeling/image_encoder.py:122-144. x,y: (B,3,H,W) -> two (B,256,H/16,W/16)."""
    pw, pb = sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"]
    ps = cfg["patch"]
    x = F.conv2d(x, pw, pb, stride=ps).permute(0, 2, 3, 1)
    y = F.conv2d(y, pw, pb, stride=ps).permute(0, 2, 3, 1)
    if (p + "pos_embed") in sd:
        pos = F.interpolate(sd[p + "pos_embed"].permute(0, 3, 1, 2), size=(x.shape[1], x.shape[2]),
                            mode="bicubic", align_corners=False).permute(0, 2, 3, 1)
        x, y = x + pos, y + pos
    blk = adapter_fusion_block if cfg["mod"] == "sam_adpt" else lora_block
    for i in range(cfg["depth"]):
        ws = 0 if i in cfg["global_idx"] else cfg["window"]
        x, y = blk(sd, f"{p}blocks.{i}.", x, y, cfg["heads"], ws)
    outs = []
    for t in (x, y):
        t = F.conv2d(t.permute(0, 3, 1, 2), sd[p + "neck.0.weight"])
        t = layer_norm_2d(sd, p + "neck.1.", t)
        t = F.conv2d(t, sd[p + "neck.2.weight"], padding=1)
        outs.append(layer_norm_2d(sd, p + "neck.3.", t))
    return outs[0], outs[1]


# --------------------------------------------------------------------------------------------
# FPN / fusion / UNetFormer decoder
# --------------------------------------------------------------------------------------------
class BNState:
    """Collects BatchNorm running-stat updates made during a training-mode forward."""
    def __init__(self, training: bool, momentum: float = 0.1):
        self.training = training
        self.momentum = momentum
        self.updates: Dict[str, torch.Tensor] = {}


def batch_norm(sd: SD, p: str, x: torch.Tensor, st: BNState, eps: float = 1e-5):
    """nn.BatchNorm2d semantics (unetformer_mmsam.py:24-73,508): batch statistics + running-stat
    update in training, running statistics in eval."""
    rm, rv = sd[p + "running_mean"], sd[p + "running_var"]
    if st.training:
        rm2, rv2 = rm.clone(), rv.clone()
        out = F.batch_norm(x, rm2, rv2, sd[p + "weight"], sd[p + "bias"], True, st.momentum, eps)
        st.updates[p + "running_mean"] = rm2
        st.updates[p + "running_var"] = rv2
        st.updates[p + "num_batches_tracked"] = sd[p + "num_batches_tracked"] + 1
        return out
    return F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], False, 0.0, eps)


def _pad_for(k: int) -> int:
    return (k - 1) // 2          # ((stride-1)+dilation*(k-1))//2 with stride=dilation=1 (:28)


def conv_bn(sd, p, x, st, k, relu6=False):
    """ConvBN / ConvBNReLU (unetformer_mmsam.py:24-40)."""
    x = F.conv2d(x, sd[p + "0.weight"], padding=_pad_for(k))
    x = batch_norm(sd, p + "1.", x, st)
    return F.relu6(x) if relu6 else x


def sep_conv_bn(sd, p, x, st, k):
    """SeparableConvBN (unetformer_mmsam.py:64-73): depthwise kxk -> BN -> 1x1."""
    C = x.shape[1]
    x = F.conv2d(x, sd[p + "0.weight"], padding=_pad_for(k), groups=C)
    x = batch_norm(sd, p + "1.", x, st)
    return F.conv2d(x, sd[p + "2.weight"])


def gla_rel_index(ws: int) -> torch.Tensor:
    """unetformer_mmsam.py:135-145: pair-wise relative position index inside a ws x ws window."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def global_local_attention(sd, p, x, st, heads: int, ws: int):
    """unetformer_mmsam.py:161-198 (GlobalLocalAttention.forward)."""
    B, C, H, W = x.shape
    local = conv_bn(sd, p + "local2.", x, st, 1) + conv_bn(sd, p + "local1.", x, st, 3)
    if W % ws:
        x = F.pad(x, (0, ws - W % ws, 0, 0), mode="reflect")      # the reference's 2-tuple form; this torch wants all 4
    if H % ws:
        x = F.pad(x, (0, 0, 0, ws - H % ws), mode="reflect")
    Hp, Wp = x.shape[2:]
    qkv = F.conv2d(x, sd[p + "qkv.0.weight"])
    d = C // heads
    hh, ww = Hp // ws, Wp // ws
    t = qkv.view(B, 3, heads, d, hh, ws, ww, ws).permute(1, 0, 4, 6, 2, 5, 7, 3)
    q, k, v = t.reshape(3, B * hh * ww, heads, ws * ws, d).unbind(0)
    dots = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    idx = sd.get(p + "relative_position_index", gla_rel_index(ws))
    bias = sd[p + "relative_position_bias_table"][idx.view(-1)].view(ws * ws, ws * ws, -1)
    dots = dots + bias.permute(2, 0, 1)[None]
    o = dots.softmax(-1) @ v                                   # (B*hh*ww, heads, ws*ws, d)
    o = o.view(B, hh, ww, heads, ws, ws, d).permute(0, 3, 6, 1, 4, 2, 5).reshape(B, C, Hp, Wp)
    o = o[:, :, :H, :W]
    out = F.avg_pool2d(F.pad(o, (0, 0, 0, 1), mode="reflect"), (ws, 1), 1, (ws // 2 - 1, 0)) + \
        F.avg_pool2d(F.pad(o, (0, 1, 0, 0), mode="reflect"), (1, ws), 1, (0, ws // 2 - 1))
    out = out + local
    out = F.pad(out, (0, 1, 0, 1), mode="reflect")
    out = sep_conv_bn(sd, p + "proj.", out, st, ws)
    return out[:, :, :H, :W]


def decoder_block(sd, p, x, st, heads, ws):
    """unetformer_mmsam.py:212-217 (Block) with Mlp :96-102 (drop=0)."""
    x = x + global_local_attention(sd, p + "attn.", batch_norm(sd, p + "norm1.", x, st), st, heads, ws)
    h = batch_norm(sd, p + "norm2.", x, st)
    h = F.conv2d(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.conv2d(F.relu6(h), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h


def _fuse(sd, p, x, res):
    """shared by WF (:242-246) and FeatureRefinementHead (:342-345)."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    w = F.relu(sd[p + "weights"])
    fw = w / (w.sum() + 1e-8)
    return fw[0] * F.conv2d(res, sd[p + "pre_conv.0.weight"]) + fw[1] * x


def wf(sd, p, x, res, st):
    """unetformer_mmsam.py:242-248."""
    return conv_bn(sd, p + "post_conv.", _fuse(sd, p, x, res), st, 3, relu6=True)


def feature_refinement_head(sd, p, x, res, st):
    """unetformer_mmsam.py:341-354."""
    x = conv_bn(sd, p + "post_conv.", _fuse(sd, p, x, res), st, 3, relu6=True)
    C = x.shape[1]
    shortcut = conv_bn(sd, p + "shortcut.", x, st, 1)
    pa = torch.sigmoid(F.conv2d(x, sd[p + "pa.0.weight"], sd[p + "pa.0.bias"], padding=1, groups=C)) * x
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.conv2d(F.relu6(F.conv2d(g, sd[p + "ca.1.0.weight"])), sd[p + "ca.3.0.weight"])
    ca = torch.sigmoid(g) * x
    x = sep_conv_bn(sd, p + "proj.", pa + ca, st, 3) + shortcut
    return F.relu6(x)


def decoder(sd, p, res1, res2, res3, res4, h, w, st, cfg, drop_mask=None):
    """unetformer_mmsam.py:444-457 (Decoder.forward). ``drop_mask`` (B,C,1,1) of {0,1/(1-p)} is the
    host-supplied Dropout2d mask for training-mode parity; None = dropout disabled."""
    heads, ws = cfg["dec_heads"], cfg["dec_window"]
    x = decoder_block(sd, p + "b4.", conv_bn(sd, p + "pre_conv.", res4, st, 1), st, heads, ws)
    x = decoder_block(sd, p + "b3.", wf(sd, p + "p3.", x, res3, st), st, heads, ws)
    x = decoder_block(sd, p + "b2.", wf(sd, p + "p2.", x, res2, st), st, heads, ws)
    x = feature_refinement_head(sd, p + "p1.", x, res1, st)
    x = conv_bn(sd, p + "segmentation_head.0.", x, st, 3, relu6=True)
    if drop_mask is not None:
        x = x * drop_mask
    x = F.conv2d(x, sd[p + "segmentation_head.2.0.weight"])
    return F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)


def squeeze_excite(sd, p, x):
    """unetformer_mmsam.py:262-266."""
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(F.conv2d(g, sd[p + "fc.0.weight"], sd[p + "fc.0.bias"]))
    g = torch.sigmoid(F.conv2d(g, sd[p + "fc.2.weight"], sd[p + "fc.2.bias"]))
    return x * g


def fpn(sd, deep, m: str, st):
    """unetformer_mmsam.py:494-516,545-552: four pyramid levels for modality ``m`` ('x'|'y')."""
    t = F.conv_transpose2d(deep, sd[f"fpn1{m}.0.weight"], sd[f"fpn1{m}.0.bias"], stride=2)
    if m == "x":
        t = F.layer_norm(t.permute(0, 2, 3, 1), (t.shape[1],), sd["fpn1x.1.ln.weight"],
                         sd["fpn1x.1.ln.bias"], 1e-6).permute(0, 3, 1, 2)
    else:
        t = batch_norm(sd, "fpn1y.1.", t, st)
    r1 = F.conv_transpose2d(F.gelu(t), sd[f"fpn1{m}.3.weight"], sd[f"fpn1{m}.3.bias"], stride=2)
    r2 = F.conv_transpose2d(deep, sd[f"fpn2{m}.0.weight"], sd[f"fpn2{m}.0.bias"], stride=2)
    return r1, r2, deep, F.max_pool2d(deep, 2, 2)


def unetformer(sd: SD, x: torch.Tensor, y: torch.Tensor, cfg: dict, training: bool = False,
               drop_mask=None, bn_state: Optional[BNState] = None):
    """unetformer_mmsam.py:533-563 (UNetFormer.forward). x (B,3,H,W), y (B,H,W) -> (B,ncls,H,W)."""
    st = bn_state if bn_state is not None else BNState(training)
    h, w = x.shape[-2:]
    y3 = y.unsqueeze(1).repeat(1, 3, 1, 1)
    dx, dy = image_encoder(sd, "image_encoder.", x, y3, cfg)
    rx = fpn(sd, dx, "x", st)
    ry = fpn(sd, dy, "y", st)
    res = [squeeze_excite(sd, f"fusion{i + 1}.se_rgb.", rx[i]) +
           squeeze_excite(sd, f"fusion{i + 1}.se_depth.", ry[i]) for i in range(4)]
    return decoder(sd, "decoder.", res[0], res[1], res[2], res[3], h, w, st, cfg, drop_mask)


# --------------------------------------------------------------------------------------------
# loss / optimiser / metrics / sliding window
# --------------------------------------------------------------------------------------------
def cross_entropy_2d(logits: torch.Tensor, target: torch.Tensor, weight=None, ignore_index: int = -100):
    """utils2.py:115-127 (CrossEntropy2d) and the ignore_index=255 variant
    train_multiscale_main.py:30-43: mean CE over (B*H*W) pixels."""
    C = logits.shape[1]
    flat = logits.permute(0, 2, 3, 1).reshape(-1, C)
    return F.cross_entropy(flat, target.reshape(-1), weight, ignore_index=ignore_index)


def sgd_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor],
             bufs: Dict[str, torch.Tensor], lr: float, momentum: float, wd: float):
    """torch.optim.SGD as configured in trainer.py:111-116 (dampening 0, no nesterov): in-place on
    params/bufs; tensors without a grad are untouched."""
    for k, g in grads.items():
        if g is None:
            continue
        d = g + wd * params[k]
        if momentum:
            if k in bufs:
                bufs[k].mul_(momentum).add_(d)
            else:
                bufs[k] = d.clone()
            d = bufs[k]
        params[k].sub_(lr * d)


def sliding_window_coords(H: int, W: int, step: int, win: Tuple[int, int]):
    """utils2.py:134-142: row-major grid with tail positions clamped to the edge (so clamped rows /
    columns are visited more than once)."""
    out = []
    for x in range(0, H, step):
        if x + win[0] > H:
            x = H - win[0]
        for y in range(0, W, step):
            if y + win[1] > W:
                y = W - win[1]
            out.append((x, y, win[0], win[1]))
    return out


def inference_windows(height: int, width: int, window: Tuple[int, int], overlap_px: int):
    """inference.py:498-533 (_get_sliding_windows): stride = window - overlap; keep tiles at least
    half a window in both dims; shift a short tail tile back to the edge."""
    wh, ww = window
    sh, sw = wh - overlap_px, ww - overlap_px
    wins = []
    for y in range(0, height, sh):
        for x in range(0, width, sw):
            ye, xe = min(y + wh, height), min(x + ww, width)
            if (ye - y) >= wh // 2 and (xe - x) >= ww // 2:
                yy, xx = y, x
                if ye == height and ye - yy < wh:
                    yy = max(0, height - wh)
                if xe == width and xe - xx < ww:
                    xx = max(0, width - ww)
                    # NB: the reference mutates its loop variable x here; later tiles of the row
                    # are generated from the loop's own range(), so there is no carry-over.
                wins.append((yy, xx, ye, xe))
    return wins


def predict_sliding(forward, img: torch.Tensor, dsm: torch.Tensor, n_classes: int, stride: int,
                    batch: int, win: Tuple[int, int]):
    """evaluator.py:113-176 (_predict_image): img (H,W,3) float, dsm (H,W); global DSM min-max,
    batched windows, SUM of raw logits in float64, argmax."""
    H, W = img.shape[:2]
    pred = torch.zeros(H, W, n_classes, dtype=torch.float64)
    d = (dsm - dsm.min()) / (dsm.max() - dsm.min())
    coords = sliding_window_coords(H, W, stride, win)
    for i in range(0, len(coords), batch):
        chunk = coords[i:i + batch]
        ip = torch.stack([img[x:x + w, y:y + h].permute(2, 0, 1) for x, y, w, h in chunk])
        dp = torch.stack([d[x:x + w, y:y + h] for x, y, w, h in chunk])
        out = forward(ip, dp)
        for o, (x, y, w, h) in zip(out, chunk):
            pred[x:x + w, y:y + h] += o.permute(1, 2, 0).double()
    return pred.argmax(-1), pred


def confusion_metrics(pred, gt, n_classes: int):
    """utils2.py:168-216 (metrics) without printing: returns dict(OA, F1[], kappa, mIoU over the
    first 5 classes)."""
    import numpy as np
    pred = np.asarray(pred).ravel()
    gt = np.asarray(gt).ravel()
    valid = (gt >= 0) & (gt < n_classes) & (pred >= 0) & (pred < n_classes)
    cm = np.bincount(gt[valid] * n_classes + pred[valid], minlength=n_classes ** 2).reshape(n_classes, n_classes)
    total = cm.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        oa = 100.0 * np.trace(cm) / float(total)
        f1 = 2.0 * np.diag(cm) / (cm.sum(1) + cm.sum(0))
        pa = np.trace(cm) / float(total)
        pe = np.sum(cm.sum(0) * cm.sum(1)) / float(total * total)
        kappa = (pa - pe) / (1 - pe)
        iou = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
    return dict(cm=cm, OA=oa, F1=f1, kappa=kappa, mIoU=float(np.nanmean(iou[:5])))


# ---------------------------------------------------------------------------------------------------------
# training input pipeline (multiscale_dataset.py:190-215, 228-245) - numpy, exactly the reference's call order
def augment_patch(data, dsm, label, hflip, vflip, rot_k):
    """data uint8 [3,S,S], dsm [S,S], label [S,S]: np.flip / np.rot90 in the reference's order (:228-245)."""
    import numpy as np
    if hflip:
        data, dsm, label = np.flip(data, axis=2).copy(), np.flip(dsm, axis=1).copy(), np.flip(label, axis=1).copy()
    if vflip:
        data, dsm, label = np.flip(data, axis=1).copy(), np.flip(dsm, axis=0).copy(), np.flip(label, axis=0).copy()
    if rot_k:
        data = np.rot90(data, k=rot_k, axes=(1, 2)).copy()
        dsm, label = np.rot90(dsm, k=rot_k, axes=(0, 1)).copy(), np.rot90(label, k=rot_k, axes=(0, 1)).copy()
    return data, dsm, label


def jitter_patch(data, brightness, contrast):
    """(:250-265) brightness / contrast jitter of a uint8 patch [3,S,S], the reference's fp32 statements in its order."""
    import numpy as np
    p = data.astype(np.float32)
    for c in range(3):
        p[c] = p[c] * brightness
        m = p[c].mean()
        p[c] = (p[c] - m) * contrast + m
        p[c] = np.clip(p[c], 0, 255)
    return p.astype(data.dtype)


def normalise_patch(data, dsm, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """(:199-208) RGB / 255 then per-channel (x - mean) / std; DSM min-max to [0,1] when max > min."""
    import numpy as np
    d = data.astype(np.float32) / 255.0
    for c in range(3):
        d[c] = (d[c] - mean[c]) / std[c]
    z = dsm.astype(np.float32)
    if z.max() > z.min():
        z = (z - z.min()) / (z.max() - z.min())
    return d.astype(np.float32), z.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------
# inference.py pre / post steps around the tile loop - numpy, the reference's statements in the reference's order
def normalize_image(image, is_rgb=True, nodata_value=None):
    """BuildingInference._normalize_image (inference.py:444-496): -> (fp32 image, bool nodata mask [H,W]).
    The percentile scalars are cast to fp32 before they meet the fp32 array, which is what numpy 1.24 (the
    reference's pin, environment.yml:19) does through value-based casting."""
    import numpy as np
    if nodata_value is not None:
        nodata_mask = np.any(image == nodata_value, axis=0) if image.ndim > 2 else (image == nodata_value)
    else:
        nodata_mask = np.zeros(image.shape[-2:], dtype=bool)
    out = image.copy().astype(np.float32)
    if is_rgb:
        if image.dtype == np.uint8:
            out = out / np.float32(255.0)
        elif image.dtype == np.uint16:
            out = out / np.float32(65535.0)
        elif np.any(~nodata_mask):
            valid = out[:, ~nodata_mask] if out.ndim > 2 else out[~nodata_mask]
            p2, p98 = np.percentile(valid, (2, 98))
            out = np.clip((out - np.float32(p2)) / np.float32(p98 - p2), 0, 1).astype(np.float32)
        else:
            out = np.clip(out, 0, 1)
        if out.ndim == 3 and out.shape[0] == 3:
            mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
            std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
            out = (out - mean) / std
    if out.ndim > 2:
        out[:, nodata_mask] = 0
    else:
        out[nodata_mask] = 0
    return out, nodata_mask


def tile_vector_masks(forward, rgb, dem, window=(256, 256), overlap_px=64, threshold=0.5, cls=1):
    """Tile loop of _tile_by_tile_vector_inference (inference.py:713-757) up to the vectoriser's input: per window
    reflect-pad (:726-737), per-tile DEM min-max (:547-555), softmax(logits)[cls] (:562-566), crop (:742-744) and
    `prediction >= threshold` (:582). rgb [3,H,W] normalised fp32 numpy, dem [H,W] raw. -> [(window, prob, mask)]"""
    import numpy as np
    out = []
    for (y, x, ye, xe) in inference_windows(rgb.shape[1], rgb.shape[2], window, overlap_px):
        wh, ww = ye - y, xe - x
        ph, pw = max(0, window[0] - wh), max(0, window[1] - ww)
        rp, dp = rgb[:, y:ye, x:xe], dem[None, y:ye, x:xe]
        if ph > 0 or pw > 0:
            rp = np.pad(rp, ((0, 0), (0, ph), (0, pw)), mode="reflect")
            dp = np.pad(dp, ((0, 0), (0, ph), (0, pw)), mode="reflect")
        d = dp[0].astype(np.float32)
        d = (d - d.min()) / (d.max() - d.min()) if d.max() > d.min() else np.zeros_like(d)
        logits = forward(torch.from_numpy(np.ascontiguousarray(rp))[None], torch.from_numpy(d)[None])
        prob = torch.softmax(logits, dim=1)[0, cls].numpy()[:wh, :ww]
        out.append(((y, x, ye, xe), prob, (prob >= threshold).astype(np.uint8)))
    return out
