"""MFNet model drop-in: `UNetFormer(decode_channels=64, dropout=0.1, window_size=8, num_classes=6)`,
`forward(x:(B,3,H,W), y:(B,H,W), mode='Train') -> (B,num_classes,H,W)`
(reference: unetformer_mmsam.py:479-563; also exported as `UNetFormer_MMSAM`).

Same attribute names (.sam, .image_encoder, .fpn{1..4}{x,y}, .fusion{1..4}, .decoder), the same
parameter names / state-dict layout (1088 keys for ViT-B MMLoRA, encoder registered under two names)
and the same freeze rule. The multimodal SAM encoder runs on the gfx950 kernels (ssrs_amd/vit.py).

The classes below are parameter containers with the reference's names; their arithmetic lives in
ssrs_amd/decoder_ops.py (one autograd.Function per module, channels-last, csrc/decoder.hip + the bf16
GEMM). No torch arithmetic op is on the forward / backward path.
"""
from __future__ import annotations

import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import decoder_ops as D
from .medsam import cfg
from .sam import sam_model_registry


# Standalone forwards of the sub-modules (Mlp, GlobalLocalAttention, SqueezeAndExcitation, SEFusion): the same kernels the
# fused Block / FpnFusion functions run, for callers that use a module on its own (feature inspection, unit tests, reuse in
# another head). They are FORWARD-ONLY - computed under no_grad; training goes through the fused autograd functions, which
# own the hand-written backward.


class Norm2d(nn.Module):
    """unetformer_mmsam.py:14-22."""

    def __init__(self, embed_dim):
        super().__init__()
        self.ln = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, x):
        return D.layer_norm_nhwc(x.permute(0, 2, 3, 1).contiguous(), self.ln.weight, self.ln.bias, self.ln.eps).permute(0, 3, 1, 2).contiguous()


def _same_pad(k, stride=1, dilation=1):
    return ((stride - 1) + dilation * (k - 1)) // 2


def _conv(cin, cout, k, groups=1, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, bias=bias, padding=_same_pad(k), groups=groups)


class ConvBNReLU(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3):
        super().__init__(_conv(cin, cout, kernel_size), nn.BatchNorm2d(cout), nn.ReLU6())


class ConvBN(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3):
        super().__init__(_conv(cin, cout, kernel_size), nn.BatchNorm2d(cout))


class Conv(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3, bias=False):
        super().__init__(_conv(cin, cout, kernel_size, bias=bias))


class SeparableConvBN(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3):
        super().__init__(_conv(cin, cin, kernel_size, groups=cin), nn.BatchNorm2d(cout), nn.Conv2d(cin, cout, 1, bias=False))


def _only(name, value, accepted, why):
    """The reference's constructor keywords are kept (unetformer_mmsam.py:86-88, 200-202); values the HIP kernels do not
    implement are refused by name instead of surfacing as a TypeError or, worse, being ignored."""
    if value not in accepted:
        raise NotImplementedError(f"{name}={value!r}: the HIP path implements {why} (what MFNet uses)")


class Mlp(nn.Module):
    """unetformer_mmsam.py:86-102: `Mlp(in_features, hidden_features=None, out_features=None, act_layer=nn.ReLU6, drop=0.)`."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.ReLU6, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        _only("act_layer", act_layer, (nn.ReLU6,), "the fused ReLU6 epilogue")
        _only("drop", float(drop), (0.0,), "drop=0")
        _only("out_features", out_features, (in_features,), "out_features == in_features (the residual form)")
        self.fc1 = nn.Conv2d(in_features, hidden_features, 1, 1, 0, bias=True)
        self.act = act_layer()
        self.fc2 = nn.Conv2d(hidden_features, out_features, 1, 1, 0, bias=True)
        self.drop = nn.Dropout(drop, inplace=True)          # p = 0: an identity kept for the attribute surface (no parameters)

    @torch.no_grad()
    def forward(self, x):
        """unetformer_mmsam.py:96-102 (forward only)."""
        B, _, H, W = x.shape
        out, _ = D.mlp_forward(self, D.to_rows(x, torch.bfloat16))
        return D.to_nchw(out, B, H, W)


class GlobalLocalAttention(nn.Module):
    """unetformer_mmsam.py:105-198."""

    def __init__(self, dim=256, num_heads=16, qkv_bias=False, window_size=8, relative_pos_embedding=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.ws = ws = window_size
        self.qkv = Conv(dim, 3 * dim, kernel_size=1, bias=qkv_bias)
        self.local1 = ConvBN(dim, dim, kernel_size=3)
        self.local2 = ConvBN(dim, dim, kernel_size=1)
        self.proj = SeparableConvBN(dim, dim, kernel_size=window_size)
        self.relative_pos_embedding = relative_pos_embedding
        if relative_pos_embedding:
            self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
            c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
            rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
            rel[:, :, 0] += ws - 1
            rel[:, :, 1] += ws - 1
            rel[:, :, 0] *= 2 * ws - 1
            self.register_buffer("relative_position_index", rel.sum(-1))
            nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    @torch.no_grad()
    def forward(self, x):
        """unetformer_mmsam.py:161-198 (forward only; BatchNorm follows self.training, running statistics included)."""
        B, _, H, W = x.shape
        out, _ = D.gla_forward(self, D.to_rows(x, torch.bfloat16), (B, H, W), self.training)
        return D.to_nchw(out, B, H, W)


class Block(nn.Module):
    """unetformer_mmsam.py:200-217, with the reference's keyword surface."""

    def __init__(self, dim=256, num_heads=16, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0., act_layer=nn.ReLU6,
                 norm_layer=nn.BatchNorm2d, window_size=8):
        super().__init__()
        _only("qkv_bias", bool(qkv_bias), (False,), "a bias-free qkv conv")
        _only("attn_drop", float(attn_drop), (0.0,), "attn_drop=0 (the reference's Block never passes it on either)")
        _only("drop_path", float(drop_path), (0.0,), "drop_path=0")
        _only("norm_layer", norm_layer, (nn.BatchNorm2d,), "BatchNorm2d")
        self.norm1 = norm_layer(dim)
        self.attn = GlobalLocalAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, window_size=window_size)
        self.drop_path = nn.Identity()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), out_features=dim, act_layer=act_layer, drop=drop)
        self.norm2 = norm_layer(dim)

    def forward_rows(self, x, geom):
        """x fp32 [B*H*W, C] channels-last rows."""
        return D.BlockFn.apply(x, self, geom, self.training, *D.block_params(self))

    def forward(self, x):
        B, C, H, W = x.shape
        return D.to_nchw(self.forward_rows(D.to_rows(x.float()), (B, H, W)), B, H, W)


class _Fuse(nn.Module):
    def __init__(self, in_channels, decode_channels, eps=1e-8):
        super().__init__()
        self.pre_conv = Conv(in_channels, decode_channels, kernel_size=1)
        self.weights = nn.Parameter(torch.ones(2, dtype=torch.float32), requires_grad=True)
        self.eps = eps
        self.post_conv = ConvBNReLU(decode_channels, decode_channels, kernel_size=3)

    _FN = None

    def forward_rows(self, x, res, perm, geom):
        """x fp32 [B*(H/2)*(W/2), C]; res bf16 [B*H*W, Cin] in row order `perm`; geom = (B, H, W)."""
        return type(self)._FN.apply(x, res, self, perm, geom, self.training, *type(self)._PARAMS(self))

    def forward(self, x, res):
        B, _, H, W = res.shape
        out = self.forward_rows(D.to_rows(x.float()), D.to_rows(res, torch.bfloat16), 0, (B, H, W))
        return D.to_nchw(out.float(), B, H, W)


class WF(_Fuse):
    """unetformer_mmsam.py:233-248."""

    def __init__(self, in_channels=128, decode_channels=128, eps=1e-8):
        super().__init__(in_channels, decode_channels, eps)

    _FN, _PARAMS = D.WFFn, staticmethod(D.wf_params)


class FeatureRefinementHead(_Fuse):
    """unetformer_mmsam.py:320-354."""

    def __init__(self, in_channels=64, decode_channels=64):
        super().__init__(in_channels, decode_channels, 1e-8)
        dc = decode_channels
        self.pa = nn.Sequential(nn.Conv2d(dc, dc, kernel_size=3, padding=1, groups=dc), nn.Sigmoid())
        self.ca = nn.Sequential(nn.AdaptiveAvgPool2d(1), Conv(dc, dc // 16, kernel_size=1), nn.ReLU6(),
                                Conv(dc // 16, dc, kernel_size=1), nn.Sigmoid())
        self.shortcut = ConvBN(dc, dc, kernel_size=1)
        self.proj = SeparableConvBN(dc, dc, kernel_size=3)
        self.act = nn.ReLU6()

    _FN, _PARAMS = D.FRHFn, staticmethod(D.frh_params)


class SqueezeAndExcitation(nn.Module):
    def __init__(self, channel, reduction=16, activation=nn.ReLU(inplace=True)):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(channel, channel // reduction, kernel_size=1), activation,
                                nn.Conv2d(channel // reduction, channel, kernel_size=1), nn.Sigmoid())

    @torch.no_grad()
    def forward(self, x):
        """unetformer_mmsam.py:262-266 (forward only): x * sigmoid(fc2(act(fc1(mean_hw(x)))))."""
        if not isinstance(self.fc[1], nn.ReLU):
            raise NotImplementedError("the HIP gate kernel implements the ReLU MFNet uses")
        B, _, H, W = x.shape
        rows = D.to_rows(x.float())
        _, _, g = D.se_gates(self, rows, B, H * W)
        out = D.ops.se_scale_sum(rows, rows, g, torch.zeros_like(g), B, H * W, out_dtype=torch.float32)
        return D.to_nchw(out, B, H, W)


class SEFusion(nn.Module):
    def __init__(self, channels_in, activation=nn.ReLU(inplace=True)):
        super().__init__()
        self.se_rgb = SqueezeAndExcitation(channels_in, activation=activation)
        self.se_depth = SqueezeAndExcitation(channels_in, activation=activation)

    @torch.no_grad()
    def forward(self, rgb, depth):
        """unetformer_mmsam.py:276-281 (forward only): se_rgb(rgb) + se_depth(depth)."""
        B, _, H, W = rgb.shape
        rx, ry = D.to_rows(rgb.float()), D.to_rows(depth.float())
        gx, gy = D.se_gates(self.se_rgb, rx, B, H * W)[2], D.se_gates(self.se_depth, ry, B, H * W)[2]
        return D.to_nchw(D.ops.se_scale_sum(rx, ry, gx, gy, B, H * W, out_dtype=torch.float32), B, H, W)


class Decoder(nn.Module):
    """unetformer_mmsam.py:419-457."""

    def __init__(self, encoder_channels=(64, 128, 256, 512), decode_channels=64, dropout=0.1, window_size=8, num_classes=6):
        super().__init__()
        dc = decode_channels
        self.pre_conv = ConvBN(encoder_channels[-1], dc, kernel_size=1)
        self.b4 = Block(dim=dc, num_heads=8, window_size=window_size)
        self.b3 = Block(dim=dc, num_heads=8, window_size=window_size)
        self.p3 = WF(encoder_channels[-2], dc)
        self.b2 = Block(dim=dc, num_heads=8, window_size=window_size)
        self.p2 = WF(encoder_channels[-3], dc)
        self.p1 = FeatureRefinementHead(encoder_channels[-4], dc)
        self.segmentation_head = nn.Sequential(ConvBNReLU(dc, dc), nn.Dropout2d(p=dropout, inplace=True),
                                               Conv(dc, num_classes, kernel_size=1))

    def forward_rows(self, res1, res2, res3, res4, B, S, h, w, perms=(2, 1)):
        """res_i: bf16 channels-last rows of the 4S, 2S, S, S/2 pyramid levels (res1 / res2 in the
        un-shuffled ConvTranspose order `perms`); S = the encoder's token grid, an int (square) or (Sh, Sw); returns logits
        (B, ncls, h, w) fp32. Any H x W whose token grid is even in both directions (H, W multiples of 32) is accepted, as the
        reference's forward does (unetformer_mmsam.py:533-563); the logits are the 16x grid, so h = 16 Sh, w = 16 Sw."""
        Sh, Sw = (S, S) if isinstance(S, int) else (int(S[0]), int(S[1]))
        if Sh % 2 or Sw % 2 or h != 16 * Sh or w != 16 * Sw:
            raise NotImplementedError("the HIP decoder handles inputs whose height and width are multiples of 32 (an even token grid, "
                                      "logits = 16x the grid); got a %dx%d grid for %dx%d logits" % (Sh, Sw, h, w))
        tr = self.training
        x = D.ConvBN1x1Fn.apply(res4, self.pre_conv, tr, *D.convbn_params(self.pre_conv))
        x = self.b4.forward_rows(x, (B, Sh // 2, Sw // 2))
        x = self.b3.forward_rows(self.p3.forward_rows(x, res3, 0, (B, Sh, Sw)), (B, Sh, Sw))
        x = self.b2.forward_rows(self.p2.forward_rows(x, res2, perms[1], (B, 2 * Sh, 2 * Sw)), (B, 2 * Sh, 2 * Sw))
        x = self.p1.forward_rows(x, res1, perms[0], (B, 4 * Sh, 4 * Sw))
        drop = self.segmentation_head[1]
        mask = None
        if drop.training and drop.p > 0:
            keep = 1.0 - drop.p
            mask = torch.bernoulli(torch.full((B, x.shape[1]), keep, device=x.device)) / keep     # Dropout2d: per (image, channel)
        return D.SegHeadFn.apply(x, self.segmentation_head, (B, 4 * Sh, 4 * Sw), 4, mask, tr, *D.seghead_params(self.segmentation_head))

    def forward(self, res1, res2, res3, res4, h, w):
        B, _, Sh, Sw = res3.shape
        r = [D.to_rows(t, torch.bfloat16) for t in (res1, res2, res3, res4)]
        return self.forward_rows(r[0], r[1], r[2], r[3], B, (Sh, Sw), h, w, perms=(0, 0))


class UNetFormer(nn.Module):
    """MFNet. Extra keyword arguments (not in the reference, all optional): `encoder` selects the SAM
    size ('vit_b' | 'vit_l' | 'vit_h'; the reference hard-codes 'vit_l' and is edited by hand,
    unetformer_mmsam.py:488-490; env MFNET_ENCODER overrides the default), `sam_checkpoint` the local
    .pth, `args` a pre-parsed cfg namespace instead of sys.argv, `precision` 'bf16' | 'fp32' (see __init__)."""

    _CKPT = {"vit_b": "weights/sam_vit_b_01ec64.pth", "vit_l": "weights/sam_vit_l_0b3195.pth",
             "vit_h": "weights/sam_vit_h_4b8939.pth", "default": "weights/sam_vit_b_01ec64.pth"}

    def __init__(self, decode_channels=64, dropout=0.1, window_size=8, num_classes=6, *, encoder=None,
                 sam_checkpoint=None, args=None, precision=None):
        super().__init__()
        # "bf16" (default): bf16 MFMA contractions with fp32 accumulation, training + inference, logits within 2e-2 of the fp32
        # reference. "fp32": every contraction on the fp32 MFMA, fp32 activations and gradients: logits within 1e-4, gradients
        # pinned against the reference's at 1e-4 (tests/test_fp32_gpu.py); slow (no bf16 MFMA, unfused epilogues) - the mode for
        # verification and for fp32 training runs. May be switched on a built model: model.precision = "fp32".
        self.precision = precision or os.environ.get("MFNET_PRECISION", "bf16")
        if self.precision not in ("bf16", "fp32"):
            raise ValueError(f"precision {self.precision!r}: 'bf16' or 'fp32'")
        if args is None:
            args = cfg.parse_args()
        encoder = encoder or os.environ.get("MFNET_ENCODER", "vit_l")
        ckpt = sam_checkpoint if sam_checkpoint is not None else self._CKPT.get(encoder)
        if ckpt is not None and not os.path.exists(ckpt):
            warnings.warn(f"SAM checkpoint {ckpt!r} not found: the encoder keeps its random initialisation "
                          "(the reference would prompt for a download here, build_sam.py:113-142)")
            ckpt = None
        self.sam = sam_model_registry[encoder](args, checkpoint=ckpt)
        self.image_encoder = self.sam.image_encoder
        self.mod = args.mod
        ec = (256, 256, 256, 256)
        self.fpn1x = nn.Sequential(nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2), Norm2d(256), nn.GELU(),
                                   nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2))
        self.fpn2x = nn.Sequential(nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2))
        self.fpn3x = nn.Identity()
        self.fpn4x = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fpn1y = nn.Sequential(nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2), nn.BatchNorm2d(256), nn.GELU(),
                                   nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2))
        self.fpn2y = nn.Sequential(nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2))
        self.fpn3y = nn.Identity()
        self.fpn4y = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fusion1 = SEFusion(ec[0])
        self.fusion2 = SEFusion(ec[1])
        self.fusion3 = SEFusion(ec[2])
        self.fusion4 = SEFusion(ec[3])
        # freeze rule (unetformer_mmsam.py:522-528; README.md:12-13 asks for a hand edit to 'Adapter'
        # when mod=sam_adpt - here the filter follows `mod`)
        token = "Adapter" if args.mod == "sam_adpt" else "lora_"
        for n, p in self.image_encoder.named_parameters():
            p.requires_grad = token in n
        self.decoder = Decoder(ec, decode_channels, dropout, window_size, num_classes)

    def forward(self, x, y, mode='Train'):
        if x.shape[-2] % 32 or x.shape[-1] % 32:
            raise NotImplementedError("the HIP path takes tiles whose height and width are multiples of 32 (an even token grid for the "
                                      "2x2 max-pool level); got %dx%d" % (x.shape[-2], x.shape[-1]))
        if self.precision == "fp32":
            if not self.training and not torch.is_grad_enabled():
                from .fp32_path import forward_fp32          # the lean inference forward (no saved tensors)
                return forward_fp32(self, x, y)
            # fp32 training (the reference trains in fp32, trainer.py:146-159): the same modules and backward passes as below with
            # every activation, gradient and contraction in fp32 (ops.FP32); each autograd node's backward runs in that mode too
            with D.ops.precision("fp32"):
                return self._forward_train(x, y)
        return self._forward_train(x, y)

    def _forward_train(self, x, y):
        h, w = x.shape[-2:]
        B = x.shape[0]
        deep = self.image_encoder.forward_stacked(x, y)             # [2B, H/16, W/16, 256] channels-last
        res1, res2, res3, res4 = D.FpnFusionFn.apply(deep, self, self.training, *D.fpn_fusion_params(self))
        return self.decoder.forward_rows(res1, res2, res3, res4, B, (deep.shape[1], deep.shape[2]), h, w)


UNetFormer_MMSAM = UNetFormer
