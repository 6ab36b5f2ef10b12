"""Block / encoder / whole-model parity of the HIP path (bf16 MFMA, fp32 accumulate) on the GPU
against the reference's golden vectors (tests/golden, generated from /root/reference) and the CPU
oracle. Tolerance from BASELINE.json's north star: 2e-2 relative for the bf16 path (relative =
max|a-b| / max|b|), argmax agreement >= 99.9 %."""
import argparse
import functools
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import golden, rel_err, rel_l2
from ssrs_amd import detfill

pytestmark = pytest.mark.gpu
BF16_TOL = 2e-2


def _args(mod):
    return argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)


def _u(key, shape):
    return torch.from_numpy(detfill.uniform(key, shape)).float()


def _block(kind, ws):
    from ssrs_amd.vit import LoraBlock
    from ssrs_amd.vit_adapter import AdapterFusionBlock
    cls = LoraBlock if kind == "lora" else AdapterFusionBlock
    return cls(args=_args("sam_lora" if kind == "lora" else "sam_adpt"), dim=128, num_heads=2, mlp_ratio=4.0,
               qkv_bias=True, norm_layer=functools.partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU,
               use_rel_pos=True, rel_pos_zero_init=True, window_size=ws, input_size=(64, 64))


@pytest.mark.parametrize("kind", ["lora", "adpt"])
@pytest.mark.parametrize("ws", [14, 0])
@pytest.mark.parametrize("S", [16, 20])
def test_block_fwd_bwd_vs_reference(kind, ws, S):
    tag = f"block_{kind}_w{ws}_s{S}"
    g = golden(tag)
    blk = _block(kind, ws)
    blk.load_state_dict(detfill.fill_state_dict(blk.state_dict(), tag + "/"))
    blk = blk.cuda()
    trainable = [k[2:] for k in g.files if k.startswith("g/")]
    for n, p in blk.named_parameters():
        p.requires_grad = n in trainable
    x = _u(tag + "/x", (1, S, S, 128)).cuda().requires_grad_()
    y = _u(tag + "/y", (1, S, S, 128)).cuda().requires_grad_()
    ox, oy = blk(x, y)
    assert rel_err(ox, g["out_x"]) < BF16_TOL and rel_err(oy, g["out_y"]) < BF16_TOL
    ((ox * _u(tag + "/gx", ox.shape).cuda()).sum() + (oy * _u(tag + "/gy", oy.shape).cuda()).sum()).backward()
    assert rel_err(x.grad, g["grad_x"]) < BF16_TOL and rel_err(y.grad, g["grad_y"]) < BF16_TOL
    worst = {}
    for n, p in blk.named_parameters():
        if n in trainable:
            assert p.grad is not None, n
            worst[n] = rel_err(p.grad, g["g/" + n])
    bad = {k: v for k, v in worst.items() if v > 3e-2}
    assert not bad, bad


def _tiny_model(mod, num_classes=6):
    from ssrs_amd.unetformer_mmsam import UNetFormer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return UNetFormer(num_classes=num_classes, encoder="vit_tiny_test", args=_args(mod))


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_encoder_vs_reference(mod):
    tag = f"encoder_tiny_{mod}"
    g = golden(tag)
    enc = _tiny_model(mod).image_encoder
    enc.load_state_dict(detfill.fill_state_dict(enc.state_dict(), tag + "/"))
    enc = enc.cuda().eval()
    x = _u(tag + "/x", (1, 3, 256, 256)).cuda()
    y = _u(tag + "/y", (1, 3, 256, 256)).cuda()
    # the golden was made with a genuine 3-channel y; the shared patch embed is linear in the input, so
    # feed it through the RGB slot of a second call (forward_stacked(rgb=y, dsm=...)) and keep the x half
    with torch.no_grad():
        a = enc.forward_stacked(x, y[:, 0])
        b = enc.forward_stacked(y, x[:, 0])
    fx = a[:1].permute(0, 3, 1, 2)
    assert rel_err(fx, g["fx"]) < BF16_TOL
    # LoRA/adapters make the y stream depend on modality-specific weights, so only fx is comparable
    # this way; fy is covered by the whole-model tests (single-channel DSM, as MFNet feeds it).
    assert torch.isfinite(b).all()


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_model_eval_logits_vs_reference(mod):
    tag = f"model_tiny_{mod}"
    g = golden(tag)
    m = _tiny_model(mod)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().eval()
    rgb, dsm, _ = detfill.tiles(tag, 2, 256, 6)
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
    assert out.shape == (2, 6, 256, 256)
    assert rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"]) < BF16_TOL
    agree = (out.argmax(1).cpu().numpy() == g["eval_argmax"]).mean()
    assert agree >= 0.999, agree


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_model_train_step_vs_reference(mod):
    """loss, every gradient (norm + leading slice) and the loss after one SGD step."""
    tag = f"model_tiny_{mod}"
    g = golden(tag)
    m = _tiny_model(mod)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().train()
    m.decoder.segmentation_head[1].p = 0.0
    rgb, dsm, tgt = detfill.tiles(tag, 2, 256, 6)
    rgb, dsm, tgt = rgb.cuda(), dsm.cuda(), tgt.cuda()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    losses = []
    for step in range(2):
        opt.zero_grad()
        out = m(rgb, dsm)
        loss = F.cross_entropy(out.permute(0, 2, 3, 1).reshape(-1, 6), tgt.view(-1))
        loss.backward()
        losses.append(loss.item())
        if step == 0:
            assert rel_err(out[:, :, ::4, ::4], g["logits_s4"]) < BF16_TOL
            names = [str(n) for n in g["grad_names"]]
            pd = dict(m.named_parameters())
            got = {n for n, p in pd.items() if p.grad is not None}
            assert got == set(names), (got ^ set(names))
            tot_ref = np.sqrt(sum(float(g["g/" + n][0]) ** 2 for n in names))
            err2 = 0.0
            for n in names:
                want = g["g/" + n]
                gn = pd[n].grad.double().flatten()
                # per-tensor: norm within 5 % (+ a floor for near-zero grads), leading slice within 5 % of its max
                assert abs(gn.norm().item() - want[0]) <= 5e-2 * want[0] + 1e-3 * tot_ref, (n, gn.norm().item(), want[0])
                sl = gn[:256].cpu().numpy()
                w = want[2:2 + sl.size]
                err2 += float(((sl - w) ** 2).sum())
                assert np.abs(sl - w).max() <= 5e-2 * max(np.abs(w).max(), 1e-3 * tot_ref), n
        opt.step()
    assert abs(losses[0] - float(g["loss0"])) < 5e-3 * float(g["loss0"])
    assert abs(losses[1] - float(g["loss1"])) < 5e-3 * float(g["loss1"])


def test_vitb_eval_logits_vs_reference():
    """BASELINE config: ViT-B MMLoRA, 256x256 tile; logits 2e-2 relative, argmax >= 99.9 %."""
    from ssrs_amd.unetformer_mmsam import UNetFormer
    tag = "model_vitb_sam_lora"
    g = golden(tag)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder="vit_b", args=_args("sam_lora"))
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().eval()
    rgb, dsm, _ = detfill.tiles(tag, 1, 256, 6)
    with torch.no_grad():
        out = m(rgb.cuda(), dsm.cuda())
    e = rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"])
    agree = (out.argmax(1).cpu().numpy() == g["eval_argmax"]).mean()
    assert e < BF16_TOL, e
    assert agree >= 0.999, agree


def test_cpu_tensors_are_rejected_loudly():
    from ssrs_amd._lib import MfnetError
    m = _tiny_model("sam_lora")
    with pytest.raises(MfnetError):
        m(torch.zeros(1, 3, 256, 256), torch.zeros(1, 256, 256))
