#!/usr/bin/env python3
"""Developer diagnostic (GPU box): stage-by-stage error of the HIP path against the oracle run in
float64 on the same device. Prints numbers, asserts nothing. Not part of the product or the tests."""
import argparse
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import rel_err, rel_l2  # noqa: E402
from oracle import mfnet_ref as R  # noqa: E402
from ssrs_amd import detfill  # noqa: E402
from ssrs_amd.unetformer_mmsam import UNetFormer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--enc", default="vit_tiny_test")
    ap.add_argument("--mod", default="sam_lora")
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    args = argparse.Namespace(mod=a.mod, mid_dim=None, image_size=1024, multimask_output=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder=a.enc, args=args)
    tag = f"diag_{a.enc}_{a.mod}"
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda()
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in m.state_dict().items()}
    if a.enc == "vit_tiny_test":
        cfg = R.vit_cfg("tiny", mod=a.mod, dim=128, depth=2, heads=2, global_idx=(1,))
    else:
        cfg = R.vit_cfg(a.enc, mod=a.mod)
    rgb, dsm, tgt = detfill.tiles(tag, a.batch, 256, 6)
    rgb, dsm = rgb.cuda(), dsm.cuda()
    B = a.batch
    enc = m.image_encoder
    with torch.no_grad():
        # residual stream after embed and after each block
        h = enc.embed(rgb, dsm)
        pw, pb = sd["image_encoder.patch_embed.proj.weight"], sd["image_encoder.patch_embed.proj.bias"]
        import torch.nn.functional as F
        y3 = dsm.double().unsqueeze(1).repeat(1, 3, 1, 1)
        ox = F.conv2d(rgb.double(), pw, pb, stride=16).permute(0, 2, 3, 1)
        oy = F.conv2d(y3, pw, pb, stride=16).permute(0, 2, 3, 1)
        pos = F.interpolate(sd["image_encoder.pos_embed"].permute(0, 3, 1, 2), size=ox.shape[1:3], mode="bicubic",
                            align_corners=False).permute(0, 2, 3, 1)
        ox, oy = ox + pos, oy + pos
        print(f"embed      x {rel_err(h[0], ox):.2e}  y {rel_err(h[1], oy):.2e}")
        blk = R.adapter_fusion_block if a.mod == "sam_adpt" else R.lora_block
        for i, b in enumerate(enc.blocks):
            h = b.forward_stacked(h)
            ws = 0 if i in cfg["global_idx"] else cfg["window"]
            ox, oy = blk(sd, f"image_encoder.blocks.{i}.", ox, oy, cfg["heads"], ws)
            print(f"block {i:2d} ws={ws:2d}  x {rel_err(h[0], ox):.2e}  y {rel_err(h[1], oy):.2e}   |h| {h.abs().max().item():.2f}")
        deep = enc.forward_stacked(rgb, dsm)
        fx, fy = R.image_encoder(sd, "image_encoder.", rgb.double(), y3, cfg)
        print(f"neck       x {rel_err(deep[:B].permute(0, 3, 1, 2), fx):.2e}  y {rel_err(deep[B:].permute(0, 3, 1, 2), fy):.2e}")
        for training in (False, True):
            m.train(training)
            m.decoder.segmentation_head[1].p = 0.0
            out = m(rgb, dsm)
            ref = R.unetformer(sd, rgb.double(), dsm.double(), cfg, training=training)
            agree = (out.argmax(1) == ref.argmax(1)).float().mean().item()
            print(f"model train={training}: logits rel {rel_err(out, ref):.2e}  argmax agree {agree:.5f}")
            # exact (float64) decoder applied to the PRODUCT's encoder features: isolates how much the
            # decoder amplifies the encoder's bf16 rounding
            def tail(dx, dy):
                st = R.BNState(training)
                rx, ry = R.fpn(sd, dx, "x", st), R.fpn(sd, dy, "y", st)
                res = [R.squeeze_excite(sd, f"fusion{i + 1}.se_rgb.", rx[i]) +
                       R.squeeze_excite(sd, f"fusion{i + 1}.se_depth.", ry[i]) for i in range(4)]
                return R.decoder(sd, "decoder.", res[0], res[1], res[2], res[3], 256, 256, st, cfg)
            px, py = deep[:B].permute(0, 3, 1, 2).double(), deep[B:].permute(0, 3, 1, 2).double()
            mixed = tail(px, py)
            print(f"   exact decoder on product features vs oracle: {rel_err(mixed, ref):.2e}; product decoder vs exact decoder on same features: {rel_err(out, mixed):.2e}")
            for lvl in (1e-3, 3e-3):
                gen = torch.Generator(device="cuda").manual_seed(1)
                nx = fx * (1 + lvl * torch.randn(fx.shape, generator=gen, device="cuda", dtype=torch.float64))
                ny = fy * (1 + lvl * torch.randn(fy.shape, generator=gen, device="cuda", dtype=torch.float64))
                print(f"   exact decoder, features perturbed by {lvl:.0e} relative gaussian noise: {rel_err(tail(nx, ny), ref):.2e}")


def grads():
    """gradient of the CE loss w.r.t. the residual stream after every block and w.r.t. the neck output,
    product vs float64 oracle, training mode."""
    import torch.nn.functional as F
    ap = argparse.ArgumentParser()
    ap.add_argument("--enc", default="vit_tiny_test")
    ap.add_argument("--mod", default="sam_lora")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--grads", action="store_true")
    ap.add_argument("--eval-bn", action="store_true")
    a = ap.parse_args()
    args = argparse.Namespace(mod=a.mod, mid_dim=None, image_size=1024, multimask_output=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = UNetFormer(num_classes=6, encoder=a.enc, args=args)
    tag = f"diag_{a.enc}_{a.mod}"
    m.load_state_dict(detfill.fill_state_dict(m.state_dict(), tag + "/"))
    m = m.cuda()
    training = not a.eval_bn
    m.train(training)
    m.decoder.segmentation_head[1].p = 0.0
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in m.state_dict().items()}
    cfg = R.vit_cfg("tiny", mod=a.mod, dim=128, depth=2, heads=2, global_idx=(1,)) if a.enc == "vit_tiny_test" else R.vit_cfg(a.enc, mod=a.mod)
    rgb, dsm, tgt = detfill.tiles(tag, a.batch, 256, 6)
    rgb, dsm, tgt = rgb.cuda(), dsm.cuda(), tgt.cuda()
    B = a.batch
    enc = m.image_encoder
    # ---- product
    hs = []
    h = enc.embed(rgb, dsm).requires_grad_()
    hs.append(h)
    for b in enc.blocks:
        h = b.forward_stacked(h)
        h.retain_grad()
        hs.append(h)
    from ssrs_amd.vit import _Neck
    two, _, Sh, Sw, D = h.shape
    deep = _Neck.apply(h.view(2 * B, Sh, Sw, D), enc)
    deep.retain_grad()
    dx_, dy_ = deep[:B].permute(0, 3, 1, 2), deep[B:].permute(0, 3, 1, 2)
    res = [getattr(m, f"fusion{i}")(getattr(m, f"fpn{i}x")(dx_), getattr(m, f"fpn{i}y")(dy_)) for i in (1, 2, 3, 4)]
    out = m.decoder(*res, 256, 256)
    loss = F.cross_entropy(out.permute(0, 2, 3, 1).reshape(-1, 6), tgt.view(-1))
    loss.backward()
    # ---- oracle
    y3 = dsm.double().unsqueeze(1).repeat(1, 3, 1, 1)
    pw, pb = sd["image_encoder.patch_embed.proj.weight"], sd["image_encoder.patch_embed.proj.bias"]
    ox = F.conv2d(rgb.double(), pw, pb, stride=16).permute(0, 2, 3, 1)
    oy = F.conv2d(y3, pw, pb, stride=16).permute(0, 2, 3, 1)
    pos = F.interpolate(sd["image_encoder.pos_embed"].permute(0, 3, 1, 2), size=ox.shape[1:3], mode="bicubic", align_corners=False).permute(0, 2, 3, 1)
    ox, oy = (ox + pos).requires_grad_(), (oy + pos).requires_grad_()
    os_ = [(ox, oy)]
    blk = R.adapter_fusion_block if a.mod == "sam_adpt" else R.lora_block
    for i in range(cfg["depth"]):
        ws = 0 if i in cfg["global_idx"] else cfg["window"]
        ox, oy = blk(sd, f"image_encoder.blocks.{i}.", ox, oy, cfg["heads"], ws)
        ox.retain_grad(); oy.retain_grad()
        os_.append((ox, oy))
    feats = []
    for t in (ox, oy):
        t = F.conv2d(t.permute(0, 3, 1, 2), sd["image_encoder.neck.0.weight"])
        t = R.layer_norm_2d(sd, "image_encoder.neck.1.", t)
        t = F.conv2d(t, sd["image_encoder.neck.2.weight"], padding=1)
        t = R.layer_norm_2d(sd, "image_encoder.neck.3.", t)
        t.retain_grad()
        feats.append(t)
    st = R.BNState(training)
    rx, ry = R.fpn(sd, feats[0], "x", st), R.fpn(sd, feats[1], "y", st)
    rs = [R.squeeze_excite(sd, f"fusion{i + 1}.se_rgb.", rx[i]) + R.squeeze_excite(sd, f"fusion{i + 1}.se_depth.", ry[i]) for i in range(4)]
    ref = R.decoder(sd, "decoder.", rs[0], rs[1], rs[2], rs[3], 256, 256, st, cfg)
    lref = F.cross_entropy(ref.permute(0, 2, 3, 1).reshape(-1, 6), tgt.view(-1))
    lref.backward()
    print(f"loss {loss.item():.6f} ref {lref.item():.6f}")
    gd = deep.grad
    print(f"d neck-out   x {rel_err(gd[:B].permute(0, 3, 1, 2), feats[0].grad):.2e}  y {rel_err(gd[B:].permute(0, 3, 1, 2), feats[1].grad):.2e}")
    for i in range(len(hs) - 1, 0, -1):
        print(f"d block{i - 1:2d}-out x {rel_err(hs[i].grad[0], os_[i][0].grad):.2e}  y {rel_err(hs[i].grad[1], os_[i][1].grad):.2e}")
    print(f"d embed      x {rel_err(hs[0].grad[0], os_[0][0].grad):.2e}  y {rel_err(hs[0].grad[1], os_[0][1].grad):.2e}")
    # exact float64 decoder + loss evaluated at the PRODUCT's features: how much of the feature-gradient
    # difference is the decoder's own sensitivity (max-pool / ReLU6 / batch-stat switches)?
    pf = [deep.detach()[:B].permute(0, 3, 1, 2).double().requires_grad_(), deep.detach()[B:].permute(0, 3, 1, 2).double().requires_grad_()]
    st2 = R.BNState(training)
    rx, ry = R.fpn(sd, pf[0], "x", st2), R.fpn(sd, pf[1], "y", st2)
    rs = [R.squeeze_excite(sd, f"fusion{i + 1}.se_rgb.", rx[i]) + R.squeeze_excite(sd, f"fusion{i + 1}.se_depth.", ry[i]) for i in range(4)]
    o2 = R.decoder(sd, "decoder.", rs[0], rs[1], rs[2], rs[3], 256, 256, st2, cfg)
    F.cross_entropy(o2.permute(0, 2, 3, 1).reshape(-1, 6), tgt.view(-1)).backward()
    print(f"exact decoder at product features: d feat vs oracle  max {rel_err(pf[0].grad, feats[0].grad):.2e} / {rel_err(pf[1].grad, feats[1].grad):.2e}"
          f"  L2 {rel_l2(pf[0].grad, feats[0].grad):.2e} / {rel_l2(pf[1].grad, feats[1].grad):.2e}")
    print(f"product decoder bwd vs exact decoder bwd at the same features: max {rel_err(gd[:B].permute(0, 3, 1, 2), pf[0].grad):.2e}  L2 {rel_l2(gd[:B].permute(0, 3, 1, 2), pf[0].grad):.2e}")
    print("same in relative L2:")
    print(f"d neck-out   x {rel_l2(gd[:B].permute(0, 3, 1, 2), feats[0].grad):.2e}  y {rel_l2(gd[B:].permute(0, 3, 1, 2), feats[1].grad):.2e}")
    for i in range(len(hs) - 1, 0, -1):
        print(f"d block{i - 1:2d}-out x {rel_l2(hs[i].grad[0], os_[i][0].grad):.2e}  y {rel_l2(hs[i].grad[1], os_[i][1].grad):.2e}")
    print(f"d embed      x {rel_l2(hs[0].grad[0], os_[0][0].grad):.2e}  y {rel_l2(hs[0].grad[1], os_[0][1].grad):.2e}")



if __name__ == "__main__" and "--grads" in sys.argv:
    grads()
    sys.exit(0)
if __name__ == "__main__":
    main()
