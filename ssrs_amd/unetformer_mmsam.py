"""MFNet model drop-in: `UNetFormer(decode_channels=64, dropout=0.1, window_size=8, num_classes=6)`,
`forward(x:(B,3,H,W), y:(B,H,W), mode='Train') -> (B,num_classes,H,W)`
(reference: unetformer_mmsam.py:479-563; also exported as `UNetFormer_MMSAM`).

Same attribute names (.sam, .image_encoder, .fpn{1..4}{x,y}, .fusion{1..4}, .decoder), the same
parameter names / state-dict layout (1088 keys for ViT-B MMLoRA, encoder registered under two names)
and the same freeze rule. The multimodal SAM encoder runs on the gfx950 kernels (ssrs_amd/vit.py).

ROUND-1 NOTE: the FPN / SE-fusion / UNetFormer decoder below (about 3 % of the forward FLOPs,
SURVEY.md section 8d) still execute as ROCm PyTorch ops on the GPU; they are being replaced by the
HBM-bound kernels of csrc/decoder.hip row by row (DESIGN.md "status").
"""
from __future__ import annotations

import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .medsam import cfg
from .sam import sam_model_registry


class Norm2d(nn.Module):
    """unetformer_mmsam.py:14-22."""

    def __init__(self, embed_dim):
        super().__init__()
        self.ln = nn.LayerNorm(embed_dim, eps=1e-6)

    def forward(self, x):
        return self.ln(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2).contiguous()


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


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Conv2d(dim, hidden, 1, 1, 0, bias=True)
        self.act = nn.ReLU6()
        self.fc2 = nn.Conv2d(hidden, dim, 1, 1, 0, bias=True)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


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

    def forward(self, x):
        B, C, H, W = x.shape
        ws, nh = self.ws, self.num_heads
        local = self.local2(x) + self.local1(x)
        if W % ws:
            x = F.pad(x, (0, ws - W % ws), mode="reflect")
        if H % ws:
            x = F.pad(x, (0, 0, 0, ws - H % ws), mode="reflect")
        Hp, Wp = x.shape[2:]
        d, hh, ww = C // nh, Hp // ws, Wp // ws
        qkv = self.qkv(x).view(B, 3, nh, d, hh, ws, ww, ws).permute(1, 0, 4, 6, 2, 5, 7, 3)
        q, k, v = qkv.reshape(3, B * hh * ww, nh, ws * ws, d).unbind(0)
        dots = (q @ k.transpose(-2, -1)) * self.scale
        if self.relative_pos_embedding:
            bias = self.relative_position_bias_table[self.relative_position_index.view(-1)]
            dots = dots + bias.view(ws * ws, ws * ws, -1).permute(2, 0, 1).unsqueeze(0)
        o = dots.softmax(dim=-1) @ v
        o = o.view(B, hh, ww, nh, ws, ws, d).permute(0, 3, 6, 1, 4, 2, 5).reshape(B, C, Hp, Wp)[:, :, :H, :W]
        out = F.avg_pool2d(F.pad(o, (0, 0, 0, 1), mode="reflect"), (ws, 1), 1, (ws // 2 - 1, 0)) + \
            F.avg_pool2d(F.pad(o, (0, 1, 0, 0), mode="reflect"), (1, ws), 1, (0, ws // 2 - 1))
        out = F.pad(out + local, (0, 1, 0, 1), mode="reflect")
        return self.proj(out)[:, :, :H, :W]


class Block(nn.Module):
    def __init__(self, dim=256, num_heads=16, mlp_ratio=4., window_size=8):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(dim)
        self.attn = GlobalLocalAttention(dim, num_heads=num_heads, qkv_bias=False, window_size=window_size)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.norm2 = nn.BatchNorm2d(dim)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _Fuse(nn.Module):
    def __init__(self, in_channels, decode_channels, eps=1e-8):
        super().__init__()
        self.pre_conv = Conv(in_channels, decode_channels, kernel_size=1)
        self.weights = nn.Parameter(torch.ones(2, dtype=torch.float32), requires_grad=True)
        self.eps = eps
        self.post_conv = ConvBNReLU(decode_channels, decode_channels, kernel_size=3)

    def fuse(self, x, res):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        w = F.relu(self.weights)
        fw = w / (torch.sum(w, dim=0) + self.eps)
        return self.post_conv(fw[0] * self.pre_conv(res) + fw[1] * x)


class WF(_Fuse):
    """unetformer_mmsam.py:233-248."""

    def __init__(self, in_channels=128, decode_channels=128, eps=1e-8):
        super().__init__(in_channels, decode_channels, eps)

    def forward(self, x, res):
        return self.fuse(x, res)


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

    def forward(self, x, res):
        x = self.fuse(x, res)
        shortcut = self.shortcut(x)
        x = self.pa(x) * x + self.ca(x) * x
        return self.act(self.proj(x) + shortcut)


class SqueezeAndExcitation(nn.Module):
    def __init__(self, channel, reduction=16, activation=nn.ReLU(inplace=True)):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(channel, channel // reduction, kernel_size=1), activation,
                                nn.Conv2d(channel // reduction, channel, kernel_size=1), nn.Sigmoid())

    def forward(self, x):
        return x * self.fc(F.adaptive_avg_pool2d(x, 1))


class SEFusion(nn.Module):
    def __init__(self, channels_in, activation=nn.ReLU(inplace=True)):
        super().__init__()
        self.se_rgb = SqueezeAndExcitation(channels_in, activation=activation)
        self.se_depth = SqueezeAndExcitation(channels_in, activation=activation)

    def forward(self, rgb, depth):
        return self.se_rgb(rgb) + self.se_depth(depth)


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

    def forward(self, res1, res2, res3, res4, h, w):
        x = self.b4(self.pre_conv(res4))
        x = self.b3(self.p3(x, res3))
        x = self.b2(self.p2(x, res2))
        x = self.p1(x, res1)
        x = self.segmentation_head(x)
        return F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)


class UNetFormer(nn.Module):
    """MFNet. Extra keyword arguments (not in the reference, all optional): `encoder` selects the SAM
    size ('vit_b' | 'vit_l' | 'vit_h'; the reference hard-codes 'vit_l' and is edited by hand,
    unetformer_mmsam.py:488-490; env MFNET_ENCODER overrides the default), `sam_checkpoint` the local
    .pth, `args` a pre-parsed cfg namespace instead of sys.argv."""

    _CKPT = {"vit_b": "weights/sam_vit_b_01ec64.pth", "vit_l": "weights/sam_vit_l_0b3195.pth",
             "vit_h": "weights/sam_vit_h_4b8939.pth", "default": "weights/sam_vit_b_01ec64.pth"}

    def __init__(self, decode_channels=64, dropout=0.1, window_size=8, num_classes=6, *, encoder=None,
                 sam_checkpoint=None, args=None):
        super().__init__()
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
        h, w = x.shape[-2:]
        B = x.shape[0]
        deep = self.image_encoder.forward_stacked(x, y)             # [2B, H/16, W/16, 256] channels-last
        deepx = deep[:B].permute(0, 3, 1, 2)
        deepy = deep[B:].permute(0, 3, 1, 2)
        res1 = self.fusion1(self.fpn1x(deepx), self.fpn1y(deepy))
        res2 = self.fusion2(self.fpn2x(deepx), self.fpn2y(deepy))
        res3 = self.fusion3(self.fpn3x(deepx), self.fpn3y(deepy))
        res4 = self.fusion4(self.fpn4x(deepx), self.fpn4y(deepy))
        return self.decoder(res1, res2, res3, res4, h, w)


UNetFormer_MMSAM = UNetFormer
