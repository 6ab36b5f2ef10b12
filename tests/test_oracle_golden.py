"""Pin the CPU oracle (oracle/mfnet_ref.py) against outputs of the REFERENCE itself
(tests/golden/*.npz, produced by tools/make_goldens.py importing /root/reference in the build
container). Weights/inputs are regenerated from the closed form in ssrs_amd/detfill.py.
Block/encoder goldens: fp32 vs fp32, 2e-5 relative (op association only). Model-level goldens
(logits, loss, every gradient, SGD-updated parameters, BatchNorm buffers) were produced with the
reference run in float64 and are matched at 1e-9: fp32 noise cannot hide a restatement error."""
import argparse
import functools
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import golden, rel_err
from oracle import mfnet_ref as R
from ssrs_amd import detfill

TOL = 2e-5


def _args(mod):
    return argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1)


def _block(kind, ws):
    from ssrs_amd.vit import LoraBlock
    from ssrs_amd.vit_adapter import AdapterFusionBlock
    cls = LoraBlock if kind == "lora" else AdapterFusionBlock
    return cls(args=_args("sam_lora" if kind == "lora" else "sam_adpt"), dim=128, num_heads=2, mlp_ratio=4.0,
               qkv_bias=True, norm_layer=functools.partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU,
               use_rel_pos=True, rel_pos_zero_init=True, window_size=ws, input_size=(64, 64))


def _u(key, shape):
    return torch.from_numpy(detfill.uniform(key, shape)).float()


@pytest.mark.parametrize("kind", ["lora", "adpt"])
@pytest.mark.parametrize("ws", [14, 0])
@pytest.mark.parametrize("S", [16, 20])
def test_block_matches_reference(kind, ws, S):
    tag = f"block_{kind}_w{ws}_s{S}"
    g = golden(tag)
    blk = _block(kind, ws)
    sd = detfill.fill_state_dict(blk.state_dict(), tag + "/")
    trainable = [k[2:] for k in g.files if k.startswith("g/")]
    for k in trainable:
        sd[k].requires_grad_()
    x = _u(tag + "/x", (1, S, S, 128)).requires_grad_()
    y = _u(tag + "/y", (1, S, S, 128)).requires_grad_()
    fn = R.lora_block if kind == "lora" else R.adapter_fusion_block
    ox, oy = fn(sd, "", x, y, 2, ws)
    assert rel_err(ox, g["out_x"]) < TOL and rel_err(oy, g["out_y"]) < TOL
    ((ox * _u(tag + "/gx", ox.shape)).sum() + (oy * _u(tag + "/gy", oy.shape)).sum()).backward()
    assert rel_err(x.grad, g["grad_x"]) < TOL and rel_err(y.grad, g["grad_y"]) < TOL
    assert len(trainable) > 0
    for k in trainable:
        assert rel_err(sd[k].grad, g["g/" + k]) < 5e-5, k


def _tiny_model(mod, num_classes=6):
    from ssrs_amd.unetformer_mmsam import UNetFormer
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return UNetFormer(num_classes=num_classes, encoder="vit_tiny_test", args=_args(mod))


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_encoder_matches_reference(mod):
    tag = f"encoder_tiny_{mod}"
    g = golden(tag)
    enc = _tiny_model(mod).image_encoder
    sd = detfill.fill_state_dict(enc.state_dict(), tag + "/")
    cfg = R.vit_cfg("tiny", mod=mod, dim=128, depth=2, heads=2, global_idx=(1,))
    trainable = [k[2:] for k in g.files if k.startswith("g/")]
    assert trainable and all(("lora_" in k) or ("Adapter" in k) for k in trainable)
    for k in trainable:
        sd[k].requires_grad_()
    y = _u(tag + "/y", (1, 1, 256, 256)).repeat(1, 3, 1, 1)     # one plane x3, as MFNet feeds the DSM
    fx, fy = R.image_encoder(sd, "", _u(tag + "/x", (1, 3, 256, 256)), y, cfg)
    assert rel_err(fx, g["fx"]) < TOL and rel_err(fy, g["fy"]) < TOL
    ((fx * _u(tag + "/gx", fx.shape)).sum() + (fy * _u(tag + "/gy", fy.shape)).sum()).backward()
    for k in trainable:
        assert rel_err(sd[k].grad, g["g/" + k]) < (2e-4 if sd[k].numel() == 1 else 5e-5), k


def _summ(t, n=256):
    f = t.detach().double().flatten()
    return np.concatenate([[f.norm().item(), f.sum().item()], f[:n].numpy()])


def _check_summary(got, want, tol, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    if got.size <= 4:        # scalar mixing weights: a cancellation-heavy full reduction in fp32
        tol = max(tol, 1e-6)
    scale = max(np.abs(want[2:]).max(), abs(want[0]) / max(1, np.sqrt(got.size)), 1e-30)
    # grads that are mathematically zero (e.g. a conv bias feeding BatchNorm) are fp32 noise ~1e-8
    assert abs(got[0] - want[0]) <= tol * abs(want[0]) + 1e-12, (what, "norm", got[0], want[0])
    assert np.abs(got[2:] - want[2:]).max() <= tol * scale + 1e-13, (what, "slice")


def _dbl(sd):
    return {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}


def run_oracle_train_steps(g, tag, mod, ncls, ignore=None, tol=1e-9):
    """Two SGD steps through the oracle, compared with the reference's recorded losses / grads /
    updated parameters / BatchNorm buffers."""
    m = _tiny_model(mod, ncls) if "tiny" in tag else None
    sd = _dbl(detfill.fill_state_dict(m.state_dict(), tag + "/"))      # model-level goldens are float64
    cfg = R.vit_cfg("tiny", mod=mod, dim=128, depth=2, heads=2, global_idx=(1,), num_classes=ncls)
    names = [str(n) for n in g["grad_names"]]
    key = lambda n: n[4:] if n.startswith("sam.image_encoder.") else n      # noqa: E731
    params = {key(n): sd[key(n)].clone().requires_grad_() for n in names}
    rgb, dsm, tgt = detfill.tiles(tag, 2, 256, ncls)
    rgb, dsm = rgb.double(), dsm.double()
    if ignore is not None:
        tgt = tgt.clone()
        tgt[:, :32, :48] = ignore
    bufs = {}
    for step in range(2):
        work = dict(sd)
        work.update(params)
        st = R.BNState(True)
        out = R.unetformer(work, rgb, dsm, cfg, training=True, bn_state=st)
        loss = R.cross_entropy_2d(out, tgt, torch.ones(ncls, dtype=torch.float64) if ignore is None else None,
                                  ignore_index=-100 if ignore is None else ignore)
        grads = torch.autograd.grad(loss, list(params.values()))
        assert abs(loss.item() - float(g[f"loss{step}"])) < tol * abs(float(g[f"loss{step}"]))
        if step == 0:
            assert rel_err(out[:, :, ::4, ::4], g["logits_s4"]) < tol
            for n, gr in zip(names, grads):
                _check_summary(_summ(gr), g["g/" + n], 20 * tol, n)
        with torch.no_grad():
            R.sgd_step({k: v for k, v in params.items()}, dict(zip(params.keys(), grads)), bufs, 0.01, 0.9, 5e-4)
            for k, v in st.updates.items():
                sd[k] = v.detach()
    for n in names:
        _check_summary(_summ(params[key(n)]), g["p/" + n], 20 * tol, "p/" + n)
    for k in g.files:
        if k.startswith("b/"):
            want = g[k]
            got = sd[k[2:]].double()
            if want.size == got.numel():
                assert rel_err(got, want.reshape(got.shape)) < tol, k
            else:
                _check_summary(_summ(got), want, tol, k)


@pytest.mark.parametrize("mod", ["sam_lora", "sam_adpt"])
def test_tiny_model_eval_and_train_steps(mod):
    tag = f"model_tiny_{mod}"
    g = golden(tag)
    m = _tiny_model(mod)
    sd = _dbl(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    cfg = R.vit_cfg("tiny", mod=mod, dim=128, depth=2, heads=2, global_idx=(1,))
    rgb, dsm, _ = detfill.tiles(tag, 2, 256, 6)
    with torch.no_grad():
        out = R.unetformer(sd, rgb.double(), dsm.double(), cfg, training=False)
    assert rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"]) < 1e-9
    assert (out.argmax(1).numpy() == g["eval_argmax"]).mean() > 0.9999
    run_oracle_train_steps(g, tag, mod, 6)


def test_tiny_model_non_square_tile():
    """The reference's forward takes any H x W (unetformer_mmsam.py:533-563): a 192 x 256 tile (12 x 16 tokens) through the imported
    reference (model_tiny_rect.npz) against the oracle restatement."""
    tag = "model_tiny_rect"
    g = golden(tag)
    m = _tiny_model("sam_lora")
    sd = _dbl(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    cfg = R.vit_cfg("tiny", mod="sam_lora", dim=128, depth=2, heads=2, global_idx=(1,))
    rgb = torch.from_numpy(0.5 * (detfill.uniform(tag + "/rgb", (2, 3, 192, 256)) + 1.0)).double()
    dsm = torch.from_numpy(0.5 * (detfill.uniform(tag + "/dsm", (2, 192, 256)) + 1.0)).double()
    with torch.no_grad():
        out = R.unetformer(sd, rgb, dsm, cfg, training=False)
    assert out.shape == (2, 6, 192, 256)
    assert rel_err(out[:, :, ::4, ::4], g["eval_logits_s4"]) < 1e-9
    assert (out.argmax(1).numpy() == g["eval_argmax"]).mean() > 0.9999


def test_tiny_model_ignore_index_255():
    tag = "model_tiny_ignore255"
    run_oracle_train_steps(golden(tag), tag, "sam_lora", 2, ignore=255)


def test_sliding_window_index_lists_and_merge():
    g = golden("sliding")
    for k in g.files:
        if k.startswith("sw_"):
            H, W, step, win = map(int, k.split("_")[1:])
            assert np.array_equal(np.array(R.sliding_window_coords(H, W, step, (win, win))), g[k]), k
        elif k.startswith("inf_"):
            H, W, ov = map(int, k.split("_")[1:])
            assert np.array_equal(np.array(R.inference_windows(H, W, (256, 256), ov)), g[k]), k
    H, W = 600, 700
    img = torch.from_numpy((0.5 * (detfill.uniform("sliding/img", (H, W, 3)) + 1)).astype(np.float32))
    dsm = torch.from_numpy((37.0 + 11.0 * detfill.uniform("sliding/dsm", (H, W))).astype(np.float32))

    def standin(ip, dp):
        a = ip.mean(1) * 2 - dp
        return torch.stack([a, dp * 0.7 + ip[:, 0] * 0.1], 1)
    pred, _ = R.predict_sliding(standin, img, dsm, 2, 64, 10, (256, 256))
    assert (pred.numpy() == g["predict_600_700_argmax"]).mean() > 0.99999


def test_metrics_match_reference():
    p = ((detfill.uniform("metrics/p", (5000,)) + 1) * 3).astype(np.int64).clip(0, 5)
    gt = ((detfill.uniform("metrics/g", (5000,)) + 1) * 3).astype(np.int64).clip(0, 5)
    gt[::3] = p[::3]
    m = R.confusion_metrics(p, gt, 6)
    assert abs(m["mIoU"] - float(golden("metrics")["miou"])) < 1e-12


# ------------------------------------------------------------------------------------------ pre / post steps (prepost.npz)
def test_normalize_image_restatement_vs_reference():
    """R.normalize_image against BuildingInference._normalize_image itself (inference.py:444-496) on the seven dtype /
    nodata cases. Integer inputs: bit-equal. The fp32 percentile-stretch cases (3, 4): the fixture was produced under
    numpy 2.2, whose promotion rules run those statements in float64 (np.float64 percentiles meet an fp32 array), while
    the restatement follows numpy 1.24 - the reference's pin - and stays in fp32: equal to fp32 rounding."""
    from golden_inputs import prepost_inputs
    g = golden("prepost")
    for k, (img, is_rgb, nd) in enumerate(prepost_inputs()):
        out, mask = R.normalize_image(img, is_rgb, nd)
        assert np.array_equal(mask, g[f"norm{k}_mask"]), k
        want = g[f"norm{k}_out"]
        assert out.dtype == np.float32 and out.shape == want.shape
        if want.dtype == np.float32:
            assert np.array_equal(out, want), (k, np.abs(out - want).max())
        else:
            assert k in (3, 4) and np.abs(out - want).max() < 2e-6, (k, np.abs(out - want).max())


def test_tile_loop_restatement_vs_reference():
    """R.tile_vector_masks against _tile_by_tile_vector_inference + _predict_patch (inference.py:687-767, 535-568): window
    list, reflect padding (twice-reflected on the raster that is window/2 high), per-tile DEM min-max with the constant-
    elevation fallback, softmax class-1 probability, crop."""
    from golden_inputs import prepost_rasters, standin_model
    g = golden("prepost")
    for name, rgb, dem in prepost_rasters():
        ref = R.tile_vector_masks(standin_model, R.normalize_image(rgb, True, None)[0], dem, (64, 64), 32, 0.5, 1)
        wins = g[f"tiles_{name}_windows"]
        assert [tuple(w) for w, _, _ in ref] == [tuple(int(v) for v in w) for w in wins], name
        for i, (_, prob, mask) in enumerate(ref):
            want = g[f"tiles_{name}_prob{i}"]
            assert prob.shape == want.shape and np.abs(prob - want).max() < 1e-6, (name, i, np.abs(prob - want).max())
            assert np.array_equal(mask[np.abs(want - 0.5) > 1e-6], (want >= 0.5)[np.abs(want - 0.5) > 1e-6])


def test_dataset_patch_pipeline_restatement_vs_reference():
    """R.augment_patch + R.normalise_patch against MultiScaleSegmentationDataset.__getitem__ / _apply_augmentation
    (multiscale_dataset.py:155-245) for all 16 flip / rot90 codes, incl. the flat-DSM patch that keeps its values."""
    from golden_inputs import multiscale_inputs
    g = golden("prepost")
    S, rgb, dsm, lab = multiscale_inputs()
    for code in range(16):
        r0, c0 = (code // 8) * S, (code % 8) * S
        a, z, l = R.augment_patch(rgb[:, r0:r0 + S, c0:c0 + S], dsm[r0:r0 + S, c0:c0 + S], lab[r0:r0 + S, c0:c0 + S],
                                  code & 1, (code >> 1) & 1, (code >> 2) & 3)
        a, z = R.normalise_patch(a, z)
        assert np.abs(a - g[f"msd{code}_rgb"]).max() < 1e-6, code
        assert np.abs(z - g[f"msd{code}_dsm"]).max() < 1e-6, code
        assert np.array_equal(l, g[f"msd{code}_lab"].astype(np.int64)), code
    assert float(g["msd8_dsm"].max()) == 7.0 and float(g["msd8_dsm"].min()) == 7.0


def test_state_dict_layout_matches_reference():
    """The drop-in's ViT-B state dict is key-for-key the reference's (1088 keys MMLoRA / 1136 MMAdapter, same order, shapes and
    dtypes; the encoder under `sam.image_encoder.*` and `image_encoder.*`, SAM's unused heads included) and a checkpoint in
    that layout loads strictly (inference.py:167-173)."""
    import argparse
    import json
    import os
    import warnings
    from helpers import GOLD
    from ssrs_amd.unetformer_mmsam import UNetFormer
    for mod, nkeys in (("sam_lora", 1088), ("sam_adpt", 1136)):
        with open(os.path.join(GOLD, f"state_dict_vitb_{mod}.json")) as f:
            want = json.load(f)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = UNetFormer(num_classes=6, encoder="vit_b", args=argparse.Namespace(mod=mod, mid_dim=None, image_size=1024, multimask_output=1))
        sd = m.state_dict()
        assert len(want) == nkeys and list(sd.keys()) == list(want.keys()), mod
        for k, v in sd.items():
            assert [list(v.shape), str(v.dtype)] == want[k], (mod, k)
        # strict load of a checkpoint written in the reference's layout
        ck = {k: torch.zeros(shape, dtype=getattr(torch, dt.split(".")[1])) for k, (shape, dt) in want.items()}
        m.load_state_dict(ck, strict=True)


def test_jitter_restatement_vs_reference():
    """R.jitter_patch against _apply_augmentation's brightness / contrast branch (multiscale_dataset.py:250-265) run on the imported
    reference with a scripted `random` (tests/golden/jitter.npz): the uint8 patch bit for bit, then the normalised patch."""
    from golden_inputs import JITTER_CASES, multiscale_inputs
    g = golden("jitter")
    S, rgb, dsm, lab = multiscale_inputs()
    changed = 0
    for i, (code, bf, cf) in enumerate(JITTER_CASES):
        pr, pc = divmod(i + 3, 8)
        raw = rgb[:, pr * S:(pr + 1) * S, pc * S:(pc + 1) * S]
        a, z, l = R.augment_patch(raw, dsm[pr * S:(pr + 1) * S, pc * S:(pc + 1) * S], lab[pr * S:(pr + 1) * S, pc * S:(pc + 1) * S],
                                  code & 1, (code >> 1) & 1, (code >> 2) & 3)
        j = R.jitter_patch(a, bf, cf)
        assert j.dtype == np.uint8 and np.array_equal(j, g[f"jit{i}_u8"]), i
        changed += int((j != a).sum())
        an, zn = R.normalise_patch(j, z)
        assert np.abs(an - g[f"jit{i}_rgb"]).max() < 1e-6 and np.abs(zn - g[f"jit{i}_dsm"]).max() < 1e-6, i
        assert np.array_equal(l, g[f"jit{i}_lab"].astype(np.int64)), i
    assert changed > 1000                        # the fixtures do exercise the branch
