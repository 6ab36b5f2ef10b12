"""Sliding-window inference and the train engine on the GPU (through the C ABI) against the oracle and
the reference's golden outputs."""
import argparse
import warnings

import numpy as np
import pytest
import torch

from helpers import golden, rel_err
from oracle import mfnet_ref as R
from ssrs_amd import detfill

pytestmark = pytest.mark.gpu


def _standin(ip, dp):
    a = ip.mean(1) * 2 - dp
    return torch.stack([a, dp * 0.7 + ip[:, 0] * 0.1], 1)


def test_evaluator_predict_matches_reference_golden():
    """Evaluator._predict_image with the stand-in model the golden was made with (evaluator.py:113-176):
    clamped tail windows, whole-raster DSM min-max, summed logits, arg-max."""
    from ssrs_amd.inference import Evaluator
    g = golden("sliding")
    H, W = 600, 700
    img = (0.5 * (detfill.uniform("sliding/img", (H, W, 3)) + 1)).astype(np.float32)
    dsm = (37.0 + 11.0 * detfill.uniform("sliding/dsm", (H, W))).astype(np.float32)

    class Net:
        def forward(self, a, b, mode="Test"):
            return _standin(a, b)
    pred = Evaluator(n_classes=2, batch_size=10)._predict_image(Net(), img, dsm, 64, 10, (256, 256))
    assert pred.shape == (H, W)
    # the merge accumulates in float64 like the reference's numpy array: exact sums, so the only pixels that may differ are those
    # where the stand-in's own fp32 logits differ in the last bit between this GPU and the CPU that wrote the golden
    mism = int((pred != g["predict_600_700_argmax"]).sum())
    print(f"\n[evaluator merge] pixels differing from the reference's mask: {mism} of {H * W}")
    assert mism == 0, mism                       # measured: bit-identical to the reference mask
    # ... and the result does not depend on the order the windows land in: bit-identical scores on a second run
    from ssrs_amd.inference import SlidingWindowPredictor
    p = SlidingWindowPredictor(lambda a, b: _standin(a, b), 2, (256, 256), batch_size=10)
    it, dt = torch.from_numpy(img).permute(2, 0, 1).contiguous().cuda(), torch.from_numpy(dsm).cuda()
    (m1, (s1, _)), (m2, (s2, _)) = p.predict(it, dt, 64, return_scores=True), p.predict(it, dt, 64, return_scores=True)
    assert s1.dtype == torch.float64 and torch.equal(s1, s2) and torch.equal(m1, m2)
    p32 = SlidingWindowPredictor(lambda a, b: _standin(a, b), 2, (256, 256), batch_size=10, accumulate=torch.float32)
    assert (p32.predict(it, dt, 64) == m1).float().mean() > 0.9999


def _tiny(ncls=6):
    from ssrs_amd.unetformer_mmsam import UNetFormer
    args = argparse.Namespace(mod="sam_lora", mid_dim=None, image_size=1024, multimask_output=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=ncls, encoder="vit_tiny_test", args=args)
    return m


def test_sliding_predict_with_model_vs_oracle():
    from ssrs_amd.inference import SlidingWindowPredictor
    m = _tiny()
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "infer/"))
    sd = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    H, W = 320, 384
    img = torch.from_numpy(0.5 * (detfill.uniform("infer/img", (3, H, W)) + 1)).float()
    dsm = torch.from_numpy(20 + 5 * detfill.uniform("infer/dsm", (H, W))).float()
    p = SlidingWindowPredictor(lambda a, b: m(a, b), 6, (256, 256), batch_size=4)
    mask, (scores, lo) = p.predict(img.cuda(), dsm.cuda(), 64, return_scores=True)
    cfg = R.vit_cfg("tiny", mod="sam_lora", dim=128, depth=2, heads=2, global_idx=(1,))
    with torch.no_grad():
        ref_mask, ref = R.predict_sliding(lambda a, b: R.unetformer(sd, a.double(), b.double(), cfg), img.permute(1, 2, 0), dsm, 6, 64, 4, (256, 256))
    assert lo == 0 and scores.shape == (6, H, W)
    assert rel_err(scores.permute(1, 2, 0), ref) < 2e-2
    agree = float((mask.cpu().numpy() == ref_mask.numpy()).mean())
    print(f"\n[sliding predict, tiny model] summed-logit argmax agreement with the oracle {100 * agree:.3f} %")
    assert agree >= 0.994, agree


def test_graph_replayed_forward_is_bit_identical_to_eager():
    """SlidingWindowPredictor / RasterInference with graph=True: the forward of each batch shape (full batches and the ragged
    tail) is captured once into a HIP graph and replayed - same kernels, same order, so the rasters must be bit-identical."""
    from ssrs_amd.inference import RasterInference, SlidingWindowPredictor
    m = _tiny(2)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "infer/"))
    m = m.cuda().eval()
    H, W = 600, 712
    img = torch.from_numpy(0.5 * (detfill.uniform("graph/img", (3, H, W)) + 1)).float().cuda()
    dsm = torch.from_numpy(20 + 5 * detfill.uniform("graph/dsm", (H, W))).float().cuda()
    fwd = lambda a, b: m(a, b)                                  # noqa: E731
    eager = SlidingWindowPredictor(fwd, 2, (256, 256), batch_size=8)
    graph = SlidingWindowPredictor(fwd, 2, (256, 256), batch_size=8, graph=True)
    me, (se, _) = eager.predict(img, dsm, 128, return_scores=True)
    for _ in range(3):                                          # eager warm-up call, capturing call, replaying call
        mg, (sg, _) = graph.predict(img, dsm, 128, return_scores=True)
        assert torch.equal(sg, se) and torch.equal(mg, me)
    assert len(graph.forward._graphs) == 2                      # 20 windows in batches of 8: shapes (8, ...) and (4, ...)
    re_, rg = RasterInference(fwd, batch_size=8), RasterInference(fwd, batch_size=8, graph=True)
    pe, qe = re_.predict(img, dsm)
    for _ in range(3):
        pg, qg = rg.predict(img, dsm)
        # the blend sums overlapping tiles as 64-bit fixed-point integers (integer atomics commute): bit equality, whatever
        # order the atomics land in
        assert torch.equal(pg, pe) and torch.equal(qg, qe)
    # the same predictor twice more, eagerly: bit-identical rasters run to run
    for _ in range(2):
        pb, qb = re_.predict(img, dsm)
        assert torch.equal(pb, pe) and torch.equal(qb, qe)


def test_raster_inference_blend_vs_numpy():
    """inference.py tile loop: window grid with shifted tail tiles, reflect padding of a short raster edge,
    per-tile DSM min-max, softmax class-1 probability; overlap blend = mean over covering tiles."""
    from ssrs_amd.inference import RasterInference
    from ssrs_amd import ops
    H, W = 383, 200                                  # W < window: every tile is reflect-padded along x
    rgb = torch.from_numpy(((detfill.uniform("rinf/rgb", (3, H, W)) + 1) * 127.5).astype(np.uint8))
    dem = torch.from_numpy(100 + 30 * detfill.uniform("rinf/dem", (H, W))).float()
    dem[:40, :] = 7.0                                # a constant-elevation tile region
    ri = RasterInference(_standin, (256, 256), overlap_ratio=0.5, batch_size=3)
    prob, mask = ri.predict(rgb.cuda(), dem.cuda())
    img = ((rgb.double() / 255.0) - torch.tensor(ops.IMAGENET_MEAN).double()[:, None, None]) / torch.tensor(ops.IMAGENET_STD).double()[:, None, None]
    psum = np.zeros((H, W)); cnt = np.zeros((H, W))
    wins = R.inference_windows(H, W, (256, 256), 128)
    assert len(wins) >= 2
    for (y, x, ye, xe) in wins:
        rp = np.pad(img[:, y:ye, x:xe].numpy(), ((0, 0), (0, 256 - (ye - y)), (0, 256 - (xe - x))), mode="reflect")
        dp = np.pad(dem[y:ye, x:xe].double().numpy(), ((0, 256 - (ye - y)), (0, 256 - (xe - x))), mode="reflect")
        dp = (dp - dp.min()) / (dp.max() - dp.min()) if dp.max() > dp.min() else np.zeros_like(dp)
        out = _standin(torch.from_numpy(rp)[None], torch.from_numpy(dp)[None])
        pr = torch.softmax(out, 1)[0, 1].numpy()[:ye - y, :xe - x]
        psum[y:ye, x:xe] += pr; cnt[y:ye, x:xe] += 1
    ref = np.where(cnt > 0, psum / np.maximum(cnt, 1), 0)
    assert np.abs(prob.cpu().numpy() - ref).max() < 1e-5
    assert (mask.cpu().numpy() == (ref >= 0.5)).mean() > 0.9999


def test_normalize_raster_is_bit_equal_to_numpy_restatement():
    """_normalize_image (inference.py:444-496): uint8 / uint16 / fp32-percentile RGB, raw DEM, nodata masks; the
    device result is bit-equal to the numpy statements (IEEE fp32 divisions on both sides)."""
    from ssrs_amd import ops
    H, W = 97, 211
    u = (detfill.uniform("nr/rgb", (3, H, W)) + 1) * 0.5
    cases = [((u * 255).astype(np.uint8), True, 0), ((u * 255).astype(np.uint8), True, None), ((u * 65535).astype(np.uint16), True, 0),
             ((u * 900 - 50).astype(np.float32), True, None), ((u * 900 - 50).astype(np.float32), True, -9999.0),
             ((u[0] * 300 + 20).astype(np.float32), False, -9999), ((u[:2] * 255).astype(np.uint8), True, 3)]
    for k, (img, is_rgb, nd) in enumerate(cases):
        img = img.copy()
        if nd is not None:
            img[..., 5:9, 10:40] = nd                                   # a nodata block (in every channel)
            if img.ndim == 3:
                img[0, 50, 60] = nd                                    # and a pixel where a single channel hits it
        ref, ref_mask = R.normalize_image(img, is_rgb, nd)
        t = torch.from_numpy(img.view(np.int16)).view(torch.uint16) if img.dtype == np.uint16 else torch.from_numpy(img)
        out, mask = ops.normalize_raster(t.cuda(), is_rgb, nd)
        assert out.dtype == torch.float32 and tuple(out.shape) == img.shape, k
        assert np.array_equal(mask.cpu().numpy(), ref_mask), k
        assert np.array_equal(out.cpu().numpy(), ref), (k, np.abs(out.cpu().numpy() - ref).max())


def test_tile_masks_for_the_vectoriser_match_reference_loop():
    """predict_tiles: the reference's per-tile outputs (cropped softmax probability and `>= threshold` mask,
    inference.py:562-566, 582, 742-744) in the reference's window order."""
    from ssrs_amd.inference import RasterInference
    H, W = 383, 200
    rgb = ((detfill.uniform("rinf/rgb", (3, H, W)) + 1) * 127.5).astype(np.uint8)
    dem = (100 + 30 * detfill.uniform("rinf/dem", (H, W))).astype(np.float32)
    dem[:40, :] = 7.0
    ri = RasterInference(_standin, (256, 256), overlap_ratio=0.5, batch_size=3, threshold=0.55)
    got = list(ri.predict_tiles(torch.from_numpy(rgb).cuda(), torch.from_numpy(dem).cuda(), want_prob=True))
    ref = R.tile_vector_masks(_standin, R.normalize_image(rgb, True, None)[0], dem, (256, 256), 128, 0.55, 1)
    assert len(got) == len(ref) >= 2
    for (wg, mg, pg), (wr, pr, mr) in zip(got, ref):
        assert tuple(wg) == tuple(wr) and mg.shape == mr.shape and mg.dtype == np.uint8
        assert np.abs(pg - pr).max() < 1e-5
        sure = np.abs(pr - 0.55) > 1e-5                                  # away from the threshold the masks are equal
        assert np.array_equal(mg[sure], mr[sure]) and (mg == mr).mean() > 0.999


def test_train_engine_two_steps_vs_reference_losses():
    """TrainEngine.step (flat buffers, HIP cross-entropy, fused SGD with momentum + weight decay) twice:
    the losses match the reference's recorded loss0 / loss1 (train_step golden), and the engine's
    on-device accuracy equals the arg-max accuracy of its own logits."""
    from ssrs_amd.trainer import TrainEngine
    tag = "model_tiny_sam_lora"
    g = golden(tag)
    m = _tiny()
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda().train()
    m.decoder.segmentation_head[1].p = 0.0
    rgb, dsm, tgt = detfill.tiles(tag, 2, 256, 6)
    rgb, dsm, tgt = rgb.cuda(), dsm.cuda(), tgt.cuda()
    eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"))
    names = set(str(n) for n in g["grad_names"])
    assert set(eng.names) == names                                    # exactly the tensors the reference updates
    l0 = TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))
    l1 = TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))
    assert abs(l0["loss"] - float(g["loss0"])) < 5e-3 * float(g["loss0"])
    assert abs(l1["loss"] - float(g["loss1"])) < 5e-3 * float(g["loss1"])
    with torch.no_grad():
        acc = (m(rgb, dsm).argmax(1) == tgt).float().mean().item() * 100
    l2 = TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))
    assert abs(l2["accuracy"] - acc) < 0.5
    # parameters are views of the flat buffer and moved
    assert eng.params[0].data_ptr() == eng.flat_p.data_ptr()


def test_train_engine_graph_mode_matches_eager():
    """the HIP-graph replay of zero_grad + forward + loss + backward gives the same training trajectory as the
    eager launches (float atomics in the reductions make it equal only to rounding)."""
    from ssrs_amd.trainer import TrainEngine
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("graph", 2, 256, 6)]
    losses = {}
    for mode in (False, True):
        m = _tiny()
        m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "graph/"))
        m = m.cuda().train()
        m.decoder.segmentation_head[1].p = 0.0
        eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=mode)
        losses[mode] = [TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))["loss"] for _ in range(6)]
        if mode:
            assert len(eng._graphs) == 1
            nbt = m.decoder.b4.norm1.num_batches_tracked.item()
            assert nbt == 6, nbt                  # capture itself executes nothing: 2 eager + 4 replayed steps
    for a, b in zip(losses[False], losses[True]):
        assert abs(a - b) < 2e-3 * abs(a), (losses[False], losses[True])
    assert losses[True][-1] < losses[True][0]


def test_train_engine_graph_mode_with_rccl_exchange():
    """The data-parallel sequence of a step on a real GPU: RCCL initialised (world size 1 - one GPU here), the engine
    still issues its bucketed all-reduces after every graph replay. Capturing the graph while RCCL's watchdog thread
    polls its events must not abort (thread-local capture mode), and the trajectory equals the single-process one."""
    import os
    import torch.distributed as dist
    from ssrs_amd.trainer import TrainEngine
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("graph", 2, 256, 6)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        losses = {}
        for alone in (False, True):
            m = _tiny()
            m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "graph/"))
            m = m.cuda().train()
            m.decoder.segmentation_head[1].p = 0.0
            eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=torch.ones(6, device="cuda"), use_graph=True,
                              exchange_when_alone=alone)
            losses[alone] = [TrainEngine.read_metrics(eng.step(rgb, dsm, tgt))["loss"] for _ in range(5)]
            assert len(eng._graphs) == 1 and not eng._works
        for a, b in zip(losses[False], losses[True]):
            assert abs(a - b) < 2e-3 * abs(a), losses
    finally:
        dist.destroy_process_group()


def test_engine_gradients_equal_plain_autograd_and_bank_is_never_stale():
    """The engine's step-scoped plumbing (parameter bank, direct gradient delivery, zero arena) changes no
    arithmetic: the flat gradient buffer after forward_backward equals the gradients plain autograd returns for
    the same model without an engine; and a forward outside the engine after optimizer steps sees the UPDATED
    weights (the bank only serves derived operands inside a step)."""
    from ssrs_amd import ops
    from ssrs_amd.trainer import TrainEngine, trainable_tensors
    rgb, dsm, tgt = [t.cuda() for t in detfill.tiles("bank", 2, 256, 6)]
    w = torch.ones(6, device="cuda")

    def make():
        m = _tiny()
        m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "bank/"))
        m = m.cuda().train()
        m.decoder.segmentation_head[1].p = 0.0
        return m
    def plain_grads():
        m_ref = make()
        named, _ = trainable_tensors(m_ref)
        out = m_ref(rgb, dsm, mode="Train").contiguous()
        acc, _ = ops.ce_loss_fwd(out, tgt, w, -100)
        out.backward(ops.ce_loss_bwd(out, tgt, acc, w, -100, 1.0))
        return {n: p.grad.clone() for n, p in named}
    # Run-to-run, plain autograd itself moves small gradients (scalars like lora_wy, biases in front of a BatchNorm
    # whose true gradient is 0) by 10-30 % on this 2-tile random model: the float-atomic order of the BatchNorm /
    # pool reductions flips ReLU6 / max-pool switches. So the whole model is compared as one
    # vector ...
    ref = plain_grads()
    m = make()
    eng = TrainEngine(m, lr=0.01, momentum=0.9, weight_decay=5e-4, class_weight=w)
    flat_ref = torch.cat([ref[n].reshape(-1) for n in eng.names])
    for _ in range(2):                                # second pass runs on the bank's persistent tensors + arena
        eng.zero_grad()
        eng.forward_backward(rgb, dsm, tgt)
        flat = torch.cat([p.grad.reshape(-1) for p in eng.params])
        cos = torch.nn.functional.cosine_similarity(flat, flat_ref, dim=0).item()
        assert cos > 0.98 and abs(flat.norm().item() / flat_ref.norm().item() - 1) < 5e-2, cos
    # ... and the plumbing is pinned exactly where the map is smooth.
    # The encoder alone is a smooth map (no BatchNorm / ReLU6 / max-pool switches): with a fixed upstream gradient
    # the delivered gradients must equal plain autograd's to reduction-order noise.
    up = torch.from_numpy(detfill.uniform("bank/up", (4, 16, 16, 256))).cuda() * 1e-2
    m_plain = make()
    m_plain.image_encoder.forward_stacked(rgb, dsm).backward(up)
    plain = {n: p.grad for n, p in m_plain.named_parameters() if p.grad is not None}
    assert len(plain) == sum(1 for n in eng.names if ".image_encoder." in "." + n)
    for _ in range(2):
        eng.zero_grad()
        with eng.bank, eng.arena:
            m.image_encoder.forward_stacked(rgb, dsm).backward(up)
        for n, p in zip(eng.names, eng.params):
            if n in plain:
                r = plain[n]
                assert (p.grad - r).abs().max().item() <= 2e-4 * r.abs().max().item() + 1e-9, n
    for _ in range(2):
        eng.step(rgb, dsm, tgt)
    # outside a step every derived operand is rebuilt from the CURRENT weights ...
    from ssrs_amd import decoder_ops as D
    from ssrs_amd.bank import ParamBank
    from ssrs_amd.vit import _mix_V
    assert ParamBank.active is None and ops.ZeroArena.current is None
    w3 = m.decoder.b4.attn.local1[0].weight
    lin = m.image_encoder.blocks[0].mlp.lin1
    fresh_w3 = w3.detach().permute(0, 2, 3, 1).reshape(w3.shape[0], -1).to(torch.bfloat16)
    fresh_V = torch.cat([lin.lora_Bx.detach(), lin.lora_By.detach()], 1)
    assert torch.equal(D._wb_w3(w3), fresh_w3) and torch.equal(_mix_V(lin), fresh_V)
    # ... inside a step the bank serves its persistent tensors, re-filled from the current weights on entry
    key_w3 = ("w3", w3.data_ptr())
    stale = eng.bank.entries[key_w3][0].clone()                     # filled at the start of the LAST step
    assert not torch.equal(stale, fresh_w3)                         # the SGD update after it did move the weight
    with eng.bank:
        assert D._wb_w3(w3).data_ptr() == eng.bank.entries[key_w3][0].data_ptr()
        assert torch.equal(D._wb_w3(w3), fresh_w3) and torch.equal(_mix_V(lin), fresh_V)
    # and a forward outside the engine agrees with a fresh model holding the same state (up to the float-atomic
    # order of the SE pools, which flips single bf16 ulps in the fused features)
    m.eval()
    with torch.no_grad():
        after = m(rgb, dsm)
    m2 = make()
    m2.load_state_dict(m.state_dict())
    m2.eval()
    with torch.no_grad():
        fresh = m2(rgb, dsm)
    assert (after - fresh).abs().max().item() <= 3e-2 * fresh.abs().max().item()


def test_forward_is_bit_reproducible():
    """eval AND train-mode forward passes give bit-identical logits run after run: no float atomic is left on
    the forward path (BatchNorm statistics and the SE / FRH pools are fixed-order reductions)."""
    m = _tiny()
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "repro/"))
    m = m.cuda()
    m.decoder.segmentation_head[1].p = 0.0
    rgb, dsm, _ = [t.cuda() for t in detfill.tiles("repro", 2, 256, 6)]
    for mode in (False, True):
        m.train(mode)
        with torch.no_grad():
            outs = [m(rgb, dsm) for _ in range(4)]
        assert all(torch.equal(outs[0], o) for o in outs[1:]), mode


def test_confusion_and_metrics():
    from ssrs_amd import ops
    from ssrs_amd.inference import metrics_from_confusion
    p = ((detfill.uniform("metrics/p", (5000,)) + 1) * 3).astype(np.int64).clip(0, 5)
    gt = ((detfill.uniform("metrics/g", (5000,)) + 1) * 3).astype(np.int64).clip(0, 5)
    gt[::3] = p[::3]
    gt[::50] = 255                                                    # ignored label
    cm = ops.confusion(torch.from_numpy(p.astype(np.uint8)).cuda(), torch.from_numpy(gt).cuda(), 6).cpu().numpy()
    ref = R.confusion_metrics(p, gt, 6)
    assert np.array_equal(cm, ref["cm"])
    assert abs(metrics_from_confusion(cm)["mIoU"] - ref["mIoU"]) < 1e-12


def test_device_input_pipeline_matches_numpy_reference_order():
    """DeviceAugment (flip / rot90 / ImageNet normalisation / per-tile DSM min-max on the GPU) against the
    numpy restatement of MultiScaleDataset.__getitem__ / _apply_augmentation, every one of the 16 codes."""
    from oracle import mfnet_ref as R
    from ssrs_amd.data import DeviceAugment, random_codes
    S, n = 64, 16
    g = np.random.default_rng(0)
    rgb = g.integers(0, 256, (n, 3, S, S), dtype=np.uint8)
    dsm = (g.random((n, S, S), dtype=np.float32) * 40 + 100).astype(np.float32)
    dsm[5] = 7.0                                                     # flat tile: max == min stays as is (:206)
    lab = g.integers(0, 6, (n, S, S), dtype=np.int64)
    codes = torch.arange(n, dtype=torch.int32)
    out = DeviceAugment()(torch.from_numpy(rgb).cuda(), torch.from_numpy(dsm).cuda(), torch.from_numpy(lab).cuda(), codes)
    for t in range(n):
        c = int(codes[t])
        a, z, l = R.augment_patch(rgb[t], dsm[t], lab[t], c & 1, (c >> 1) & 1, (c >> 2) & 3)
        a, z = R.normalise_patch(a, z)
        assert np.abs(out[0][t].cpu().numpy() - a).max() < 2e-6
        assert np.abs(out[1][t].cpu().numpy() - z).max() < 1e-6
        assert np.array_equal(out[2][t].cpu().numpy(), l)
    c2 = random_codes(4096, torch.Generator().manual_seed(1))
    assert c2.dtype == torch.int32 and int(c2.max()) < 16 and abs(((c2 >> 2) == 0).float().mean().item() - 0.5) < 0.05


# ------------------------------------------------------------------------------------------ pre / post steps vs the REFERENCE's outputs
def test_normalize_raster_vs_reference_golden():
    """mfnet_normalize_raster against BuildingInference._normalize_image itself (prepost.npz): bit-equal for the integer
    rasters; the fp32 percentile-stretch cases to fp32 rounding (the fixture ran those statements in float64 under numpy 2.2)."""
    from golden_inputs import prepost_inputs
    from ssrs_amd import ops
    g = golden("prepost")
    for k, (img, is_rgb, nd) in enumerate(prepost_inputs()):
        t = torch.from_numpy(img.view(np.int16)).view(torch.uint16) if img.dtype == np.uint16 else torch.from_numpy(img)
        out, mask = ops.normalize_raster(t.cuda(), is_rgb, nd)
        want = g[f"norm{k}_out"]
        assert np.array_equal(mask.cpu().numpy(), g[f"norm{k}_mask"]), k
        if want.dtype == np.float32:
            assert np.array_equal(out.cpu().numpy(), want), (k, np.abs(out.cpu().numpy() - want).max())
        else:
            assert np.abs(out.cpu().numpy() - want).max() < 2e-6, k


def test_tile_probabilities_vs_reference_golden():
    """RasterInference.predict_tiles against the per-tile probabilities the reference's own tile loop hands to its vectoriser
    (_tile_by_tile_vector_inference + _predict_patch, prepost.npz), incl. the raster that is window/2 high (reflected twice)
    and the constant-elevation tile."""
    from golden_inputs import prepost_rasters, standin_model
    from ssrs_amd.inference import RasterInference
    g = golden("prepost")
    for name, rgb, dem in prepost_rasters():
        ri = RasterInference(standin_model, (64, 64), overlap_ratio=0.5, batch_size=5, threshold=0.5)
        got = list(ri.predict_tiles(torch.from_numpy(rgb).cuda(), torch.from_numpy(dem).cuda(), want_prob=True))
        wins = g[f"tiles_{name}_windows"]
        assert [tuple(w) for w, _, _ in got] == [tuple(int(v) for v in w) for w in wins], name
        for i, (_, mask, prob) in enumerate(got):
            want = g[f"tiles_{name}_prob{i}"]
            assert prob.shape == want.shape and np.abs(prob - want).max() < 1e-5, (name, i, np.abs(prob - want).max())
            sure = np.abs(want - 0.5) > 1e-5
            assert np.array_equal(mask[sure], (want >= 0.5)[sure].astype(np.uint8))


def test_extract_tiles_rejects_windows_that_leave_the_raster():
    from ssrs_amd import ops
    from ssrs_amd._lib import MfnetError
    from ssrs_amd.inference import _coords4
    with pytest.raises(ValueError):
        _coords4([(0, 0, 33, 10)], 32, 50, 64, 64, "cpu")
    rgb, dsm = torch.zeros(3, 32, 50, device="cuda"), torch.zeros(32, 50, device="cuda")
    with pytest.raises(MfnetError):
        ops.extract_tiles(rgb, dsm, torch.tensor([[0, 0, 33, 10]], dtype=torch.int32, device="cuda"), 64, 64, check_extents=True)


def test_device_input_pipeline_vs_reference_golden():
    """DeviceAugment against MultiScaleSegmentationDataset.__getitem__ itself (prepost.npz): the 16 flip / rot90 codes, ImageNet
    normalisation, per-patch DSM min-max with the flat patch left as it is."""
    from golden_inputs import multiscale_inputs
    from ssrs_amd.data import DeviceAugment
    g = golden("prepost")
    S, rgb, dsm, lab = multiscale_inputs()
    tiles = [(rgb[:, (c // 8) * S:(c // 8) * S + S, (c % 8) * S:(c % 8) * S + S], dsm[(c // 8) * S:(c // 8) * S + S, (c % 8) * S:(c % 8) * S + S],
              lab[(c // 8) * S:(c // 8) * S + S, (c % 8) * S:(c % 8) * S + S]) for c in range(16)]
    r = torch.from_numpy(np.stack([t[0] for t in tiles])).cuda()
    d = torch.from_numpy(np.stack([t[1] for t in tiles])).cuda()
    l = torch.from_numpy(np.stack([t[2] for t in tiles])).cuda()
    out = DeviceAugment()(r, d, l, torch.arange(16, dtype=torch.int32))
    for c in range(16):
        assert np.abs(out[0][c].cpu().numpy() - g[f"msd{c}_rgb"]).max() < 2e-6, c
        assert np.abs(out[1][c].cpu().numpy() - g[f"msd{c}_dsm"]).max() < 1e-6, c
        assert np.array_equal(out[2][c].cpu().numpy(), g[f"msd{c}_lab"].astype(np.int64)), c


def test_device_jitter_vs_reference_golden():
    """mfnet_jitter_tiles / DeviceAugment(jitter=...) against the reference's own brightness / contrast branch
    (multiscale_dataset.py:250-265; tests/golden/jitter.npz). The kernel sums the fp32 products exactly (in double), numpy adds
    them pairwise in fp32: the channel means differ in their last bits, which moves a pixel that lands within ~1e-5 of an integer
    by one grey level - so: never more than one level, on fewer than 0.1 % of the pixels (measured: printed), everything else
    bit-equal; a tile with brightness 0 is left untouched; the input batch is not modified."""
    from golden_inputs import JITTER_CASES, multiscale_inputs
    from ssrs_amd import ops
    from ssrs_amd.data import DeviceAugment
    g = golden("jitter")
    S, rgb, dsm, lab = multiscale_inputs()
    n = len(JITTER_CASES)
    sl = [divmod(i + 3, 8) for i in range(n)]
    r = torch.from_numpy(np.stack([rgb[:, a * S:(a + 1) * S, b * S:(b + 1) * S] for a, b in sl] + [rgb[:, :S, :S]])).cuda()
    d = torch.from_numpy(np.stack([dsm[a * S:(a + 1) * S, b * S:(b + 1) * S] for a, b in sl] + [dsm[:S, :S]])).cuda()
    l = torch.from_numpy(np.stack([lab[a * S:(a + 1) * S, b * S:(b + 1) * S] for a, b in sl] + [lab[:S, :S]])).cuda()
    codes = torch.tensor([c for c, _, _ in JITTER_CASES] + [0], dtype=torch.int32)
    jit = torch.tensor([[b, c] for _, b, c in JITTER_CASES] + [[0.0, 0.0]], dtype=torch.float32)
    keep = r.clone()
    # the kernel alone on the un-flipped patches: compare with the golden un-flipped (the jitter commutes with the pixel permutation)
    ju8 = ops.jitter_tiles(r.clone(), jit.cuda())
    assert torch.equal(ju8[n], r[n])                                        # brightness 0: untouched
    out = DeviceAugment()(r, d, l, codes, jitter=jit)
    assert torch.equal(r, keep)
    off = tot = 0
    for i in range(n):
        got = out[0][i].cpu().numpy()
        want = g[f"jit{i}_rgb"]
        lvl = np.abs(got - want) * 255.0 * np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]      # difference in grey levels
        assert lvl.max() < 1.0 + 1e-3, (i, lvl.max())
        off += int((lvl > 1e-3).sum()); tot += lvl.size
        assert np.abs(out[1][i].cpu().numpy() - g[f"jit{i}_dsm"]).max() < 1e-6, i
        assert np.array_equal(out[2][i].cpu().numpy(), g[f"jit{i}_lab"].astype(np.int64)), i
    print(f"\n[device jitter] {off} of {tot} pixels one grey level away from the reference")
    assert off < 1e-3 * tot
    assert (ju8[:n] != r[:n]).sum() > 1000


def test_full_size_raster_8192_properties():
    """BASELINE configs[3] at its full size (an 8192 x 8192 raster, ViT-B MMLoRA) through both predictors, held to properties that do
    not need a reference run of that size: on a non-overlapping grid every window's mask is the arg-max of a direct forward of
    that window (one contribution per pixel; >= 99.5 % of the pixels - the rest are near-ties of a differently tiled GEMM); the blended probability raster of the inference.py grid is
    bit-identical between two runs (int64 fixed-point blend) and within 2e-2 between batch sizes; its mask is the threshold of its
    probability."""
    from ssrs_amd.inference import RasterInference, SlidingWindowPredictor
    from ssrs_amd.unetformer_mmsam import UNetFormer
    args = argparse.Namespace(mod="sam_lora", mid_dim=None, image_size=1024, multimask_output=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=2, encoder="vit_b", args=args)
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), "raster8192/"))
    m = m.cuda().eval()
    S = 8192
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand((3, S, S), device="cuda", generator=g)
    dsm = 30.0 + 20.0 * torch.rand((S, S), device="cuda", generator=g)
    fwd = lambda a, b: m(a, b)      # noqa: E731
    p = SlidingWindowPredictor(fwd, 2, (256, 256), batch_size=32, graph=True)
    mask = p.predict(img, dsm, 256)                                  # 1024 windows, no overlap
    assert mask.shape == (S, S) and mask.dtype == torch.uint8
    mm = dsm.min(), dsm.max()
    d = (dsm - mm[0]) / (mm[1] - mm[0])
    with torch.no_grad():
        for (r, c) in ((0, 0), (7936, 7936), (256, 4096), (5120, 1280), (7936, 0)):
            direct = m(img[None, :, r:r + 256, c:c + 256].contiguous(), d[None, r:r + 256, c:c + 256].contiguous()).argmax(1)[0]
            # (a one-window forward takes other GEMM tile shapes than the 32-window batches: bf16-level differences may flip near-ties)
            agree = float((direct.to(torch.uint8) == mask[r:r + 256, c:c + 256]).float().mean())
            assert agree >= 0.995, (r, c, agree)
    rgb8 = (img * 255).to(torch.uint8)
    ri = RasterInference(fwd, (256, 256), overlap_ratio=0.25, batch_size=32, graph=True)
    prob1, mask1 = ri.predict(rgb8, dsm)
    prob2, mask2 = ri.predict(rgb8, dsm)
    assert torch.equal(prob1, prob2) and torch.equal(mask1, mask2)
    ri16 = RasterInference(fwd, (256, 256), overlap_ratio=0.25, batch_size=16)
    prob3, mask3 = ri16.predict(rgb8[:, :2048, :2048].contiguous(), dsm[:2048, :2048].contiguous())
    ri32 = RasterInference(fwd, (256, 256), overlap_ratio=0.25, batch_size=32)
    prob4, mask4 = ri32.predict(rgb8[:, :2048, :2048].contiguous(), dsm[:2048, :2048].contiguous())
    # other batch sizes take other GEMM tile shapes: the probabilities agree to the bf16 bound, the masks away from the threshold
    assert float((prob3 - prob4).abs().max()) < 2e-2 and float((mask3 == mask4).float().mean()) > 0.99
    assert torch.equal(mask1, (prob1 >= 0.5).to(torch.uint8))
    assert 0.0 <= float(prob1.min()) and float(prob1.max()) <= 1.0


def test_prefetcher_delivers_every_batch_intact_and_in_order():
    """ssrs_amd.data.Prefetcher (pinned double-buffered host -> device feeder of the train loop, train_multiscale_main.py:507-522):
    nine distinct batches through two pinned slots, a busy consumer stream in between (the copy of batch i + 2 must not overwrite
    the pinned buffer batch i is still being copied from), ragged last batch, and the DeviceAugment transform applied on the way."""
    from ssrs_amd.data import DeviceAugment, Prefetcher
    g = torch.Generator().manual_seed(11)
    batches = []
    for i in range(9):
        n = 4 if i < 8 else 2
        batches.append((torch.randint(0, 256, (n, 3, 64, 64), dtype=torch.uint8, generator=g),
                        100 + 20 * torch.rand((n, 64, 64), generator=g), torch.randint(0, 6, (n, 64, 64), generator=g)))
    busy = torch.rand(4096, 4096, device="cuda")
    seen = 0
    for i, (rgb, dsm, tgt) in enumerate(Prefetcher(iter(batches), "cuda")):
        busy = busy @ busy * 1e-4                                    # keeps the consumer stream behind the copies
        assert rgb.is_cuda and rgb.dtype == torch.uint8 and tgt.dtype == torch.int64
        assert torch.equal(rgb.cpu(), batches[i][0]) and torch.equal(dsm.cpu(), batches[i][1]) and torch.equal(tgt.cpu(), batches[i][2]), i
        seen += 1
    assert seen == 9
    aug = DeviceAugment(augment=False)
    outs = list(Prefetcher(iter(batches[:3]), "cuda", transform=aug))
    for i, (rgb, dsm, tgt) in enumerate(outs):
        want = aug(batches[i][0].cuda(), batches[i][1].cuda(), batches[i][2].cuda())
        assert rgb.dtype == torch.float32 and torch.equal(rgb, want[0]) and torch.equal(dsm, want[1]) and torch.equal(tgt, want[2])
