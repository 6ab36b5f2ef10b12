"""Standalone forwards of the sub-modules a user of the reference can call on their own (lora.MergedLinear, Adapter,
MLPBlock, Mlp, GlobalLocalAttention, SqueezeAndExcitation, SEFusion) and ImageEncoderViT.forward with a true 3-channel
second modality: each through the HIP kernels (C ABI), against the float64 oracle of the same reference module on the
same weights. Tolerance: the contractions run on bf16 MFMA with fp32 accumulation -> 2e-2 of the output's max (the north
star's bf16 bound); the SE gates are fp32 end to end -> 1e-5."""
import argparse
import functools
import warnings

import pytest
import torch
import torch.nn as nn

from helpers import rel_err
from oracle import mfnet_ref as R
from ssrs_amd import detfill

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _u(key, shape, scale=1.0):
    return torch.from_numpy(detfill.uniform(key, shape)).float() * scale


def _args(mod):
    return argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)


def _filled(module, tag):
    module.load_state_dict(detfill.fill_state_dict(module.state_dict(), tag + "/"))
    sd = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in module.state_dict().items()}
    return module.cuda(), sd


def _model(mod, tag):
    from ssrs_amd.unetformer_mmsam import UNetFormer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder="vit_tiny_test", args=_args(mod))
    return _filled(m, tag)


@pytest.mark.parametrize("M", [512, 400])      # tokens per modality: any multiple of 16 (a 16-row tile never straddles the modalities)
def test_merged_linear_forward(M):
    from ssrs_amd.vit import MergedLinear
    lin, sd = _filled(MergedLinear(128, 384, bias=True, r=4, enable_lora=[True, False, True]), "sa_merged")
    x, y = _u("sa_merged/x", (1, M, 128)), _u("sa_merged/y", (1, M, 128))
    ox, oy = lin(x.cuda(), y.cuda())
    assert ox.shape == (1, M, 384)
    assert rel_err(ox, R.merged_linear_qv(sd, "", x.double(), "x")) < TOL
    assert rel_err(oy, R.merged_linear_qv(sd, "", y.double(), "y")) < TOL
    # the LoRA update is really there: the result differs from the plain dense layer by more than the tolerance
    plain = torch.nn.functional.linear(x.double(), sd["weight"], sd["bias"])
    assert rel_err(ox, plain) > 2 * TOL


@pytest.mark.parametrize("skip", [True, False])
def test_adapter_forward(skip):
    from ssrs_amd.vit import Adapter
    ad, sd = _filled(Adapter(128, skip_connect=skip), f"sa_adapter{int(skip)}")
    x = _u("sa_adapter/x", (2, 16, 16, 128))
    assert rel_err(ad(x.cuda()), R.adapter(sd, "", x.double(), skip)) < TOL


def test_mlp_blocks_forward():
    from ssrs_amd import vit, vit_adapter
    x, y = _u("sa_mlp/x", (1, 16, 16, 128)), _u("sa_mlp/y", (1, 16, 16, 128))
    mlp, sd = _filled(vit.MLPBlock(128, 512, nn.GELU, lora_rank=4), "sa_mlp_lora")
    ox, oy = mlp(x.cuda(), y.cuda())
    hx, hy = R.lora_linear_mix(sd, "lin1.", x.double(), y.double())
    rx, ry = R.lora_linear_mix(sd, "lin2.", torch.nn.functional.gelu(hx), torch.nn.functional.gelu(hy))
    assert rel_err(ox, rx) < TOL and rel_err(oy, ry) < TOL
    mlp, sd = _filled(vit_adapter.MLPBlock(128, 512, nn.GELU), "sa_mlp_adpt")
    F = torch.nn.functional
    ref = F.linear(F.gelu(F.linear(x.double(), sd["lin1.weight"], sd["lin1.bias"])), sd["lin2.weight"], sd["lin2.bias"])
    assert rel_err(mlp(x.cuda()), ref) < TOL


def test_decoder_mlp_forward():
    m, sd = _model("sam_lora", "sa_dec_mlp")
    x = _u("sa_dec_mlp/x", (2, 64, 12, 12), 1.5)
    F = torch.nn.functional
    p = "decoder.b3.mlp."
    ref = F.conv2d(F.relu6(F.conv2d(x.double(), sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    assert rel_err(m.decoder.b3.mlp(x.cuda()), ref) < TOL


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("S", [16, 12])
def test_global_local_attention_forward(S, training):
    m, sd = _model("sam_lora", "sa_gla")
    attn = m.decoder.b2.attn.train(training)
    x = _u("sa_gla/x", (2, 64, S, S), 1.5)
    st = R.BNState(training)
    ref = R.global_local_attention(sd, "decoder.b2.attn.", x.double(), st, 8, 8)
    assert rel_err(attn(x.cuda()), ref) < TOL
    if training:
        assert st.updates
        for k, v in st.updates.items():
            assert rel_err(m.state_dict()[k], v) < (3e-3 if v.dtype.is_floating_point else 1e-12), k


def test_se_and_sefusion_forward():
    m, sd = _model("sam_lora", "sa_se")
    rgb, dep = _u("sa_se/rgb", (2, 256, 8, 8), 1.5), _u("sa_se/dep", (2, 256, 8, 8), 1.5)
    fu = m.fusion2
    rx = R.squeeze_excite(sd, "fusion2.se_rgb.", rgb.double())
    ry = R.squeeze_excite(sd, "fusion2.se_depth.", dep.double())
    assert rel_err(fu.se_rgb(rgb.cuda()), rx) < 1e-5
    assert rel_err(fu.se_depth(dep.cuda()), ry) < 1e-5
    assert rel_err(fu(rgb.cuda(), dep.cuda()), rx + ry) < 1e-5


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_image_encoder_forward_three_channel_y(mod):
    """ImageEncoderViT.forward(x, y) with y a real 3-channel image (image_encoder.py:122-144): the second modality's
    channels differ, so replicating one plane would give a visibly different answer (checked)."""
    m, sd = _model(mod, f"sa_enc3_{mod}")
    enc = m.image_encoder.eval()
    x, y = _u("sa_enc3/x", (1, 3, 256, 256)), _u("sa_enc3/y", (1, 3, 256, 256))
    cfg = R.vit_cfg("tiny", mod=mod, dim=128, depth=2, heads=2, global_idx=(1,))
    esd = {k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}
    with torch.no_grad():
        rx, ry = R.image_encoder(esd, "", x.double(), y.double(), cfg)
        fx, fy = enc(x.cuda(), y.cuda())
        _, fy1 = enc(x.cuda(), y[:, :1].repeat(1, 3, 1, 1).cuda())
    assert rel_err(fx, rx) < TOL and rel_err(fy, ry) < TOL
    assert rel_err(fy1, ry) > 5 * TOL
