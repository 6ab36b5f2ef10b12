"""Closed-form INPUTS of the pre / post goldens (tests/golden/prepost.npz), shared by the generator
(tools/make_goldens.py, which runs the reference on them) and the tests (which run the oracle and the HIP path on them), so
the fixture holds expected outputs only."""
import numpy as np
import torch

from ssrs_amd import detfill


def prepost_inputs():
    """Inputs of the pre/post goldens (closed form, shared with the tests): (normalize cases, raster rgb u8, raster dem)."""
    H, W = 41, 67
    u = (detfill.uniform("nrg/rgb", (3, H, W)) + 1) * 0.5
    cases = [((u * 255).astype(np.uint8), True, 0), ((u * 255).astype(np.uint8), True, None), ((u * 65535).astype(np.uint16), True, 0),
             ((u * 900 - 50).astype(np.float32), True, None), ((u * 900 - 50).astype(np.float32), True, -9999.0),
             ((u[0] * 300 + 20).astype(np.float32), False, -9999), ((u[:2] * 255).astype(np.uint8), True, 3)]
    out = []
    for img, is_rgb, nd in cases:
        img = img.copy()
        if nd is not None:
            img[..., 5:9, 10:40] = nd
            if img.ndim == 3:
                img[0, 30, 60] = nd
        out.append((img, is_rgb, nd))
    return out


def prepost_rasters():
    """[(name, rgb uint8 [3,H,W], dem fp32 [H,W])] for the tile-loop goldens; window 64, overlap 32 px. The second raster is
    exactly window/2 high: its tile is reflect-padded by its own height (np.pad reflects twice on the last row)."""
    out = []
    for name, (H, W) in (("a", (95, 50)), ("b", (32, 50)), ("c", (130, 150))):
        rgb = ((detfill.uniform(f"rinfg/{name}/rgb", (3, H, W)) + 1) * 127.5).astype(np.uint8)
        dem = (100 + 30 * detfill.uniform(f"rinfg/{name}/dem", (H, W))).astype(np.float32)
        if name == "a":
            dem[:10, :] = 7.0
        if name == "c":
            dem[:64, :64] = 3.5                  # one whole tile of constant elevation (:553-555)
        out.append((name, rgb, dem))
    return out


def standin_model(ip, dp):
    a = ip.mean(1) * 2 - dp
    return torch.stack([a, dp * 0.7 + ip[:, 0] * 0.1], 1)


def multiscale_inputs():
    S = 32
    rgb = ((detfill.uniform("msd/rgb", (3, 256, 256)) + 1) * 127.5).astype(np.uint8)
    dsm = (120 + 40 * detfill.uniform("msd/dsm", (256, 256))).astype(np.float32)
    dsm[32:64, 0:32] = 7.0                       # patch index 8 (row 1, col 0 of the 8x8 patch grid) is flat (:206)
    lab = ((detfill.uniform("msd/lab", (256, 256)) + 1) * 3).astype(np.int64).clip(0, 5)
    return S, rgb, dsm, lab


# (flip / rot90 code, brightness, contrast) of the jitter goldens (tests/golden/jitter.npz); case i acts on patch i + 3 of the 8 x 8 grid
JITTER_CASES = [(0, 1.2, 1.2), (5, 0.8, 0.8), (3, 1.2, 0.8), (12, 0.8, 1.2), (9, 1.013, 0.957), (0, 1.0, 1.0)]
