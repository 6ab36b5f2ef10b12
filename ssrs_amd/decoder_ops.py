"""FPN / SE-fusion / UNetFormer decoder on the gfx950 kernels (csrc/decoder.hip + the bf16 GEMM).

Everything runs on channels-last maps flattened to [rows, C]; each reference module is one
torch.autograd.Function with a hand-written backward (reference: unetformer_mmsam.py, cited per class).

Layout trick for the FPN: ConvTranspose2d(k=2, s=2) is a GEMM with N = 4*C whose output row m=(b,y,x)
holds the four sub-pixels (dy,dx) of that input pixel. Nothing downstream of the FPN needs the pixels in
raster order until the WF / FRH fusion (LayerNorm / BatchNorm / GELU are per pixel or per channel, the
SE pool is a sum over all pixels, the decoder's pre_conv is 1x1), so the pixel-shuffle is never run: the
fusion kernel reads its skip operand through the "un-shuffled" row order (perm = number of stacked
ConvTransposes).

Precision: bf16 MFMA operands with fp32 accumulation for every contraction, fp32 for BatchNorm
statistics, residual streams, depthwise convs, window attention and pooling.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .bank import deliver, derived, weight_bf16
from .ops import ACT_GELU, ACT_NONE, ACT_RELU6

BF = torch.bfloat16
F32 = torch.float32


# ---------------------------------------------------------------------------------------------- helpers
def _wb(w2d: torch.Tensor) -> torch.Tensor:
    """bf16 GEMM image of a (strided view of a) trainable weight: refreshed once per step by the engine's
    parameter bank, built on the spot otherwise (bank.py)."""
    return weight_bf16(w2d)


def _derived_bf16(kind, w, dst_shape, dst_view_shape, src_view):
    return derived((kind, w.data_ptr()), dst_shape, BF, lambda t: [(t.view(dst_view_shape), src_view)], device=w.device)


def _wb_w3(w):       # [O, I, 3, 3] -> bf16 [O, 9*I], k = tap*I + i (im2col3x3 order)
    w = w.detach()
    O, I = w.shape[:2]
    return _derived_bf16("w3", w, (O, 9 * I), (O, 3, 3, I), w.permute(0, 2, 3, 1))


def _wb_w3T(w):      # -> bf16 [9*I, O] for the dgrad GEMM
    w = w.detach()
    O, I = w.shape[:2]
    return _derived_bf16("w3T", w, (9 * I, O), (3, 3, I, O), w.permute(2, 3, 1, 0))


def _w3_grad(g, w):  # [O, 9*I] -> view shaped [O, I, 3, 3]
    O, I = w.shape[0], w.shape[1]
    return g.view(O, 3, 3, I).permute(0, 3, 1, 2)


def _bn(x, bn: nn.BatchNorm2d, training, act=ACT_NONE, out_dtype=F32, out2_dtype=None, chan_scale=None, rpi=0):
    if training and bn.momentum is None:
        raise NotImplementedError("BatchNorm with cumulative moving average (momentum=None) is not used by MFNet")
    return ops.bn_forward(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked if training else None,
                          training, bn.eps, bn.momentum if bn.momentum is not None else 0.1, act, chan_scale, rpi, out_dtype, out2_dtype)


def _bn_bwd(dy, x, mean, rstd, bn, training, act=ACT_NONE, dx_dtype=F32, chan_scale=None, rpi=0):
    return ops.bn_backward(dy, x, mean, rstd, bn.weight, bn.bias, training, act, chan_scale, rpi, dx_dtype)


def _wb_w3flipT(w):  # -> bf16 [I, 9*O], k = (8 - tap)*O + o: the weights of the data gradient as a 3x3 convolution of dy
    w = w.detach()
    O, I = w.shape[:2]
    src = w.reshape(O, I, 9)
    return derived(("w3fT", w.data_ptr()), (I, 9 * O), BF, lambda t: [(t.view(I, 9, O)[:, 8 - k], src[:, :, k].t()) for k in range(9)], device=w.device)


def _conv3_fwd(x_bf, w, B, H, W):
    """-> (conv output fp32 [M, O], what the backward needs). Channels that are multiples of 64 (every 3x3 conv of MFNet) take the
    implicit-GEMM kernel and keep only x; others go through the column tensor."""
    if ops.FP32 or (x_bf.shape[1] % 64 == 0 and w.shape[0] % 64 == 0):      # (fp32 training: ops.conv3x3 is im2col + the fp32 GEMM, any width)
        return ops.conv3x3(x_bf, _wb_w3(w), B, H, W), x_bf
    cols = ops.im2col3x3(x_bf, B, H, W)
    return ops.gemm(cols, _wb_w3(w), out_dtype=F32), cols


def _conv3_bwd(dy_bf, saved, w, B, H, W):
    """-> (dx fp32 [M, I], dW like w)."""
    if saved.shape[1] == w.shape[1]:                     # the input map itself: implicit-GEMM path
        dx = ops.conv3x3(dy_bf, _wb_w3flipT(w), B, H, W)
        return dx, _w3_grad(ops.conv3x3_wgrad(dy_bf, saved, B, H, W), w)
    dcols = ops.gemm(dy_bf, _wb_w3T(w))
    dx = ops.col2im3x3(dcols, B, H, W, w.shape[1])
    return dx, _w3_grad(ops.gemm_tn(dy_bf, saved, lazy=True), w)


def _w2(w):          # 1x1 conv weight [O, I, 1, 1] -> [O, I]
    return w.detach().reshape(w.shape[0], w.shape[1])


def _grads(plist, gd):
    return deliver(plist, [gd.get(id(p)) for p in plist])


def _put(gd, p, g):
    """g: a view shaped like p, or something that reshapes to it without a copy"""
    assert id(p) not in gd
    gd[id(p)] = g if g.shape == p.shape else g.view(p.shape)


# ---------------------------------------------------------------------------------------------- FPN + SEFusion
def _wb_ct(w):       # ConvTranspose2d weight [Ci, Co, 2, 2] -> bf16 GEMM weight [(dy,dx,co), ci]
    w = w.detach()
    Ci, Co = w.shape[:2]
    return _derived_bf16("ct", w, (4 * Co, Ci), (2, 2, Co, Ci), w.permute(2, 3, 1, 0))


def _wb_ctT(w):      # -> bf16 [ci, (dy,dx,co)] for the dgrad GEMM
    w = w.detach()
    Ci, Co = w.shape[:2]
    return _derived_bf16("ctT", w, (Ci, 4 * Co), (Ci, 2, 2, Co), w.permute(0, 2, 3, 1))


def _ct_grad(g, w):  # [(dy,dx,co), ci] -> view shaped [Ci, Co, 2, 2]
    return g.view(2, 2, w.shape[1], w.shape[0]).permute(3, 2, 0, 1)


def _rep4(b):        # ConvTranspose bias repeated over the four sub-pixels
    b = b.detach()
    C = b.shape[0]
    return derived(("rep4", b.data_ptr()), (4 * C,), F32, lambda t: [(t.view(4, C), b.expand(4, C))], device=b.device)


def fpn_fusion_params(model):
    ps = []
    for m in "xy":
        f1, f2 = getattr(model, f"fpn1{m}"), getattr(model, f"fpn2{m}")
        norm = f1[1].ln if m == "x" else f1[1]
        ps += [f1[0].weight, f1[0].bias, norm.weight, norm.bias, f1[3].weight, f1[3].bias, f2[0].weight, f2[0].bias]
    for i in (1, 2, 3, 4):
        fu = getattr(model, f"fusion{i}")
        for se in (fu.se_rgb, fu.se_depth):
            ps += [se.fc[0].weight, se.fc[0].bias, se.fc[2].weight, se.fc[2].bias]
    return ps


@ops.keeps_precision
class FpnFusionFn(torch.autograd.Function):
    """deep [2B, h, w, 256] fp32 (RGB images first) -> res1 (perm 2), res2 (perm 1), res3, res4, all bf16
    [rows, 256]: fpn1..4{x,y} (unetformer_mmsam.py:494-516,545-552) + fusion1..4 (:251-281,553-556)."""

    @staticmethod
    def forward(ctx, deep, model, training, *plist):
        twoB, h, w, C = deep.shape
        B = twoB // 2
        M3 = B * h * w
        d2 = deep.reshape(2 * M3, C).contiguous()
        db = ops.cast(d2, BF)
        sv = {"B": B, "h": h, "w": w, "db": db, "training": training}
        r = {}
        for mi, m in enumerate("xy"):
            f1, f2 = getattr(model, f"fpn1{m}"), getattr(model, f"fpn2{m}")
            dm = db[mi * M3:(mi + 1) * M3]
            t1 = ops.gemm(dm, _wb_ct(f1[0].weight), bias=_rep4(f1[0].bias), out_dtype=F32).view(4 * M3, C)
            if m == "x":
                n1, mean, rstd = ops.layernorm_fwd(t1, f1[1].ln.weight, f1[1].ln.bias, f1[1].ln.eps, out_dtype=BF)
            else:
                n1, _, mean, rstd = _bn(t1, f1[1], training, out_dtype=BF)
            a1 = ops.act(n1, ACT_GELU)
            r1 = ops.gemm(a1, _wb_ct(f1[3].weight), bias=_rep4(f1[3].bias)).view(16 * M3, C)
            r2 = ops.gemm(dm, _wb_ct(f2[0].weight), bias=_rep4(f2[0].bias)).view(4 * M3, C)
            r3 = d2[mi * M3:(mi + 1) * M3]
            r4, arg = ops.maxpool2_fwd(r3, B, h, w)
            r[m] = (r1, r2, r3, r4)
            sv[m] = dict(t1=t1, n1=n1, mean=mean, rstd=rstd, a1=a1, arg=arg)
        outs = []
        P = (16 * h * w, 4 * h * w, h * w, (h // 2) * (w // 2))
        for i in range(4):
            fu = getattr(model, f"fusion{i + 1}")
            gate = {}
            for m, se in (("x", fu.se_rgb), ("y", fu.se_depth)):
                gate[m] = se_gates(se, r[m][i], B, P[i])
            outs.append(ops.se_scale_sum(r["x"][i], r["y"][i], gate["x"][2], gate["y"][2], B, P[i]))
            sv[f"f{i}"] = gate
        sv["r"], sv["P"] = r, P
        ctx.sv, ctx.model, ctx.plist = sv, model, plist
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dres):
        sv, model, plist = ctx.sv, ctx.model, ctx.plist
        B, h, w, db, training, r, P = sv["B"], sv["h"], sv["w"], sv["db"], sv["training"], sv["r"], sv["P"]
        M3 = B * h * w
        C = db.shape[1]
        gd = {}
        ddeep = torch.empty((2 * M3, C), dtype=F32, device=db.device)
        dr = {"x": [None] * 4, "y": [None] * 4}
        for i in range(4):
            fu = getattr(model, f"fusion{i + 1}")
            gate = sv[f"f{i}"]
            d = dres[i].contiguous()
            dp = {}
            for m, se in (("x", fu.se_rgb), ("y", fu.se_depth)):
                s, z, g = gate[m]
                dg = ops.coldot(d, r[m][i], B, P[i])
                dpool, dW1, db1, dW2, db2 = ops.se_gate_bwd(dg, s, P[i], _w2(se.fc[0].weight), _w2(se.fc[2].weight), False, z, g, True)
                _put(gd, se.fc[0].weight, dW1); _put(gd, se.fc[0].bias, db1)
                _put(gd, se.fc[2].weight, dW2); _put(gd, se.fc[2].bias, db2)
                dp[m] = dpool
            if i == 2:      # fpn3 = identity: its gradient initialises d(deep)
                ops.se_bwd_apply(d, gate["x"][2], gate["y"][2], dp["x"], dp["y"], B, P[i], out_x=ddeep[:M3], out_y=ddeep[M3:])
            else:
                dt = F32 if i == 3 else BF
                dr["x"][i], dr["y"][i] = ops.se_bwd_apply(d, gate["x"][2], gate["y"][2], dp["x"], dp["y"], B, P[i], dt, dt)
        for mi, m in enumerate("xy"):
            f1, f2 = getattr(model, f"fpn1{m}"), getattr(model, f"fpn2{m}")
            s = sv[m]
            dm = db[mi * M3:(mi + 1) * M3]
            dd = ddeep[mi * M3:(mi + 1) * M3]
            ops.maxpool2_bwd(dr[m][3], s["arg"], dd, B, h, w, accumulate=True)
            # fpn2
            dY = dr[m][1].view(M3, 4 * C)
            ops.gemm(dY, _wb_ctT(f2[0].weight), residual=dd, out=dd)
            dw, dbias = _tn_taps(dY, dm, C)
            _put(gd, f2[0].weight, _ct_grad(dw, f2[0].weight)); _put(gd, f2[0].bias, dbias)
            # fpn1: second ConvT -> GELU -> norm -> first ConvT
            dY2 = dr[m][0].view(4 * M3, 4 * C)
            dn1 = ops.gemm(dY2, _wb_ctT(f1[3].weight), act=2, aux=s["n1"])
            dw, dbias = _tn_taps(dY2, s["a1"], C)
            _put(gd, f1[3].weight, _ct_grad(dw, f1[3].weight)); _put(gd, f1[3].bias, dbias)
            if m == "x":
                ln = f1[1].ln
                dgam = ops.zeros(C, db.device)
                dbet = ops.zeros(C, db.device)
                _, dt1 = ops.layernorm_bwd(dn1, s["t1"], s["mean"], s["rstd"], ln.weight, want_f32=False, want_bf16=True, dgamma=dgam, dbeta=dbet)
                _put(gd, ln.weight, dgam); _put(gd, ln.bias, dbet)
            else:
                dt1, dgam, dbet = _bn_bwd(dn1, s["t1"], s["mean"], s["rstd"], f1[1], training, dx_dtype=BF)
                _put(gd, f1[1].weight, dgam); _put(gd, f1[1].bias, dbet)
            dY1 = dt1.view(M3, 4 * C)
            ops.gemm(dY1, _wb_ctT(f1[0].weight), residual=dd, out=dd)
            dw, dbias = _tn_taps(dY1, dm, C)
            _put(gd, f1[0].weight, _ct_grad(dw, f1[0].weight)); _put(gd, f1[0].bias, dbias)
        return (ddeep.view(2 * B, h, w, C), None, None) + _grads(plist, gd)


def _tn_taps(dY, x, C):
    """weight gradient of a 2x2 / stride-2 transposed conv as one TN product over dY [M, 4C] (4 taps side by side), with the
    bias gradient - the column sums of dY folded over the 4 taps - from the same launch."""
    dw, dbias = ops.gemm_tn(dY, x, lazy=True, want_bias=True)
    return dw, ops.SplitSum(dbias.part.view(-1, C))


# ---------------------------------------------------------------------------------------------- ConvBN 1x1 (decoder.pre_conv)
def convbn_params(seq):
    return [seq[0].weight, seq[1].weight, seq[1].bias]


@ops.keeps_precision
class ConvBN1x1Fn(torch.autograd.Function):
    """ConvBN(kernel 1) (unetformer_mmsam.py:43-51): x bf16 [M, Cin] -> fp32 [M, Cout]."""

    @staticmethod
    def forward(ctx, x, seq, training, *plist):
        c = ops.gemm(x, _wb(_w2(seq[0].weight)), out_dtype=F32)
        y, _, mean, rstd = _bn(c, seq[1], training)
        ctx.save_for_backward(x, c, mean, rstd)
        ctx.seq, ctx.training, ctx.plist = seq, training, plist
        return y

    @staticmethod
    def backward(ctx, dy):
        x, c, mean, rstd = ctx.saved_tensors
        seq = ctx.seq
        dc, dg, dbt = _bn_bwd(dy.contiguous(), c, mean, rstd, seq[1], ctx.training, dx_dtype=BF)
        gd = {}
        _put(gd, seq[0].weight, ops.gemm_tn(dc, x, lazy=True)); _put(gd, seq[1].weight, dg); _put(gd, seq[1].bias, dbt)
        dx = ops.gemm(dc, _wb(_w2(seq[0].weight).t()))
        return (dx, None, None) + _grads(ctx.plist, gd)


# ---------------------------------------------------------------------------------------------- Block (GLA + Mlp)
def block_params(blk):
    a = blk.attn
    return [blk.norm1.weight, blk.norm1.bias, a.qkv[0].weight, a.local1[0].weight, a.local1[1].weight, a.local1[1].bias,
            a.local2[0].weight, a.local2[1].weight, a.local2[1].bias, a.proj[0].weight, a.proj[1].weight, a.proj[1].bias,
            a.proj[2].weight, a.relative_position_bias_table, blk.norm2.weight, blk.norm2.bias,
            blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias]


def _reflect_index(n, npad, device):
    i = torch.arange(npad, device=device)
    return torch.where(i < n, i, 2 * n - 2 - i)


def _reflect_rows(t, B, H, W, Hp, Wp):
    """channels-last rows [B*H*W, C] -> [B*Hp*Wp, C], reflect-padded at the bottom / right like F.pad(mode='reflect')."""
    C = t.shape[-1]
    v = t.view(B, H, W, C)
    if Hp != H:
        v = v.index_select(1, _reflect_index(H, Hp, t.device))
    if Wp != W:
        v = v.index_select(2, _reflect_index(W, Wp, t.device))
    return v.contiguous().view(B * Hp * Wp, C)


def _reflect_rows_bwd(dp, B, H, W, Hp, Wp):
    """adjoint of _reflect_rows: the padded rows' gradients are added onto the rows they mirror (fp32 sums, bf16 result)."""
    C = dp.shape[-1]
    v = dp.view(B, Hp, Wp, C).float()
    if Wp != W:
        v = torch.zeros((B, Hp, W, C), dtype=torch.float32, device=dp.device).index_add_(2, _reflect_index(W, Wp, dp.device), v)
    if Hp != H:
        v = torch.zeros((B, H, W, C), dtype=torch.float32, device=dp.device).index_add_(1, _reflect_index(H, Hp, dp.device), v)
    return v.to(dp.dtype).contiguous().view(B * H * W, C)


def gla_forward(a, n1b, geom, training, residual=None):
    """GlobalLocalAttention.forward (unetformer_mmsam.py:161-198) on bf16 channels-last rows n1b [B*H*W, C]:
    local = BN(conv3x3) + BN(conv1x1); 8x8-window attention with the relative-position bias on the (reflect-padded) map;
    avg-pool (ws,1) + (1,ws) of it; + local; reflect-pad by one, depthwise ws x ws, BN, 1x1 (+ residual).
    -> (out fp32 [B*H*W, C], tensors the backward needs)."""
    B, H, W = geom
    heads, ws = a.num_heads, a.ws
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    if Hp - H >= H or Wp - W >= W:
        raise ValueError(f"GlobalLocalAttention: a {H}x{W} map cannot be reflect-padded to the {ws}x{ws} window grid "
                         "(F.pad(mode='reflect') needs pad < size, unetformer_mmsam.py:149-155)")
    c1, cols = _conv3_fwd(n1b, a.local1[0].weight, B, H, W)
    l1, _, ml1, rl1 = _bn(c1, a.local1[1], training)
    c2 = ops.gemm(n1b, _wb(_w2(a.local2[0].weight)), out_dtype=F32)
    l2, _, ml2, rl2 = _bn(c2, a.local2[1], training)
    qkv = ops.gemm(n1b, _wb(_w2(a.qkv[0].weight)))
    table = a.relative_position_bias_table.detach().contiguous()
    if (Hp, Wp) == (H, W):
        o = ops.gla_attn_fwd(qkv, table, B, H, W, heads, a.scale)
    else:
        # pad() branch (:149-155, :166): the map is reflect-padded to whole windows BEFORE the 1x1 qkv conv; the conv is
        # pointwise, so padding its output rows is the same thing. Row gather (data movement only), window attention on
        # the padded grid, crop (:187).
        qkv_p = _reflect_rows(qkv, B, H, W, Hp, Wp)
        o = ops.gla_attn_fwd(qkv_p, table, B, Hp, Wp, heads, a.scale)
        o = o.view(B, Hp, Wp, -1)[:, :H, :W].contiguous().view(B * H * W, -1)
    s = ops.gla_pool_fwd(o, l1, l2, B, H, W)
    k = a.proj[0].weight.shape[-1]
    dwo = ops.dwconv_fwd(s, a.proj[0].weight.detach().reshape(-1, k * k), None, B, H, W, k, True)
    pb, _, mp, rp = _bn(dwo, a.proj[1], training, out_dtype=BF)
    out = ops.gemm(pb, _wb(_w2(a.proj[2].weight)), residual=residual, out_dtype=F32)
    return out, (cols, c1, ml1, rl1, c2, ml2, rl2, qkv, s, dwo, mp, rp, pb)


def mlp_forward(mlp, n2b, residual=None):
    """Mlp (unetformer_mmsam.py:86-102, drop=0): fc2(ReLU6(fc1(.))) as two GEMMs with fused bias / activation."""
    hid = ops.gemm(n2b, _wb(_w2(mlp.fc1.weight)), bias=mlp.fc1.bias, act=3)
    return ops.gemm(hid, _wb(_w2(mlp.fc2.weight)), bias=mlp.fc2.bias, residual=residual, out_dtype=F32), hid


def se_gates(se, rows, B, P):
    """SqueezeAndExcitation's per-(image, channel) gate (unetformer_mmsam.py:251-266): sigmoid(fc2(ReLU(fc1(mean))))."""
    s, _ = ops.colsum(rows, B, P)
    z, g = ops.se_gate_fwd(s, P, _w2(se.fc[0].weight), se.fc[0].bias, _w2(se.fc[2].weight), se.fc[2].bias, False)
    return s, z, g


@ops.keeps_precision
class BlockFn(torch.autograd.Function):
    """Block (unetformer_mmsam.py:200-217) = x + GlobalLocalAttention(BN(x)) (:105-198), then + Mlp(BN(.))
    (:86-102). x fp32 [B*H*W, C] -> fp32."""

    @staticmethod
    def forward(ctx, x, blk, geom, training, *plist):
        B, H, W = geom
        a = blk.attn
        x = x.contiguous()
        n1b, _, m1, r1 = _bn(x, blk.norm1, training, out_dtype=BF)
        x1, (cols, c1, ml1, rl1, c2, ml2, rl2, qkv, s, dwo, mp, rp, pb) = gla_forward(a, n1b, geom, training, residual=x)
        n2b, _, m2, r2 = _bn(x1, blk.norm2, training, out_dtype=BF)
        x2, hid = mlp_forward(blk.mlp, n2b, residual=x1)
        ctx.save_for_backward(x, m1, r1, n1b, cols, c1, ml1, rl1, c2, ml2, rl2, qkv, s, dwo, mp, rp, pb, x1, m2, r2, n2b, hid)
        ctx.blk, ctx.geom, ctx.training, ctx.plist = blk, geom, training, plist
        return x2

    @staticmethod
    def backward(ctx, dout):
        (x, m1, r1, n1b, cols, c1, ml1, rl1, c2, ml2, rl2, qkv, s, dwo, mp, rp, pb, x1, m2, r2, n2b, hid) = ctx.saved_tensors
        blk, (B, H, W), training = ctx.blk, ctx.geom, ctx.training
        a, mlp = blk.attn, blk.mlp
        gd = {}
        dout = dout.contiguous()
        d_bf = ops.cast(dout, BF)
        # ---- Mlp branch
        dhid = ops.gemm(d_bf, _wb(_w2(mlp.fc2.weight).t()), act=4, aux=hid)
        dw, dbias = ops.gemm_tn(d_bf, hid, lazy=True, want_bias=True)
        _put(gd, mlp.fc2.weight, dw); _put(gd, mlp.fc2.bias, dbias)
        dn2 = ops.gemm(dhid, _wb(_w2(mlp.fc1.weight).t()))
        dw, dbias = ops.gemm_tn(dhid, n2b, lazy=True, want_bias=True)
        _put(gd, mlp.fc1.weight, dw); _put(gd, mlp.fc1.bias, dbias)
        dx1b, dg, dbt = _bn_bwd(dn2, x1, m2, r2, blk.norm2, training)
        _put(gd, blk.norm2.weight, dg); _put(gd, blk.norm2.bias, dbt)
        dx1 = ops.add3(dout, dx1b)
        # ---- attention branch: x1 = x + proj1x1(BN(dw8x8(pool(o) + local)))
        dx1_bf = ops.cast(dx1, BF)
        dpb = ops.gemm(dx1_bf, _wb(_w2(a.proj[2].weight).t()))
        _put(gd, a.proj[2].weight, ops.gemm_tn(dx1_bf, pb, lazy=True))
        ddwo, dg, dbt = _bn_bwd(dpb, dwo, mp, rp, a.proj[1], training)
        _put(gd, a.proj[1].weight, dg); _put(gd, a.proj[1].bias, dbt)
        k = a.proj[0].weight.shape[-1]
        ds, dwdw, _ = ops.dwconv_bwd(ddwo, s, a.proj[0].weight.detach().reshape(-1, k * k), B, H, W, k, True)
        _put(gd, a.proj[0].weight, dwdw)
        d_o = ops.gla_pool_bwd(ds, B, H, W)
        table = a.relative_position_bias_table.detach().contiguous()
        ws = a.ws
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        if (Hp, Wp) == (H, W):
            dqkv, dtable = ops.gla_attn_bwd(qkv, table, d_o, B, H, W, a.num_heads, a.scale)
        else:
            C = d_o.shape[-1]
            d_op = torch.zeros((B, Hp, Wp, C), dtype=d_o.dtype, device=d_o.device)          # cropped rows receive no gradient
            d_op[:, :H, :W] = d_o.view(B, H, W, C)
            dqkv_p, dtable = ops.gla_attn_bwd(_reflect_rows(qkv, B, H, W, Hp, Wp), table, d_op.view(-1, C), B, Hp, Wp, a.num_heads, a.scale)
            dqkv = _reflect_rows_bwd(dqkv_p, B, H, W, Hp, Wp)
        _put(gd, a.relative_position_bias_table, dtable)
        dc1, dg, dbt = _bn_bwd(ds, c1, ml1, rl1, a.local1[1], training, dx_dtype=BF)
        _put(gd, a.local1[1].weight, dg); _put(gd, a.local1[1].bias, dbt)
        dc2, dg, dbt = _bn_bwd(ds, c2, ml2, rl2, a.local2[1], training, dx_dtype=BF)
        _put(gd, a.local2[1].weight, dg); _put(gd, a.local2[1].bias, dbt)
        dn1, dW3 = _conv3_bwd(dc1, cols, a.local1[0].weight, B, H, W)
        _put(gd, a.local1[0].weight, dW3)
        ops.gemm(dqkv, _wb(_w2(a.qkv[0].weight).t()), residual=dn1, out=dn1)
        ops.gemm(dc2, _wb(_w2(a.local2[0].weight).t()), residual=dn1, out=dn1)
        _put(gd, a.qkv[0].weight, ops.gemm_tn(dqkv, n1b, lazy=True)); _put(gd, a.local2[0].weight, ops.gemm_tn(dc2, n1b, lazy=True))
        dxb, dg, dbt = _bn_bwd(dn1, x, m1, r1, blk.norm1, training)
        _put(gd, blk.norm1.weight, dg); _put(gd, blk.norm1.bias, dbt)
        dx = ops.add3(dx1, dxb)
        return (dx, None, None, None) + _grads(ctx.plist, gd)


# ---------------------------------------------------------------------------------------------- WF / FRH
def wf_params(m):
    return [m.pre_conv[0].weight, m.weights, m.post_conv[0].weight, m.post_conv[1].weight, m.post_conv[1].bias]


def _fuse_fwd(x, res, perm, m, geom, training, out2_dtype=None):
    B, H, W = geom
    skip = ops.gemm(res, _wb(_w2(m.pre_conv[0].weight)))
    f = ops.upfuse_fwd(x, skip, perm, m.weights.detach(), m.eps, B, H, W)
    c, cols = _conv3_fwd(f, m.post_conv[0].weight, B, H, W)
    y, y2, mean, rstd = _bn(c, m.post_conv[1], training, act=ACT_RELU6, out2_dtype=out2_dtype)
    return y, y2, (skip, cols, c, mean, rstd)


def _fuse_bwd(dy, x, res, perm, m, geom, training, saved, gd):
    B, H, W = geom
    skip, cols, c, mean, rstd = saved
    dc, dg, dbt = _bn_bwd(dy, c, mean, rstd, m.post_conv[1], training, act=ACT_RELU6, dx_dtype=BF)
    _put(gd, m.post_conv[1].weight, dg); _put(gd, m.post_conv[1].bias, dbt)
    df, dW3 = _conv3_bwd(dc, cols, m.post_conv[0].weight, B, H, W)
    _put(gd, m.post_conv[0].weight, dW3)
    dx, dskip, dw = ops.upfuse_bwd(df, x, skip, perm, m.weights.detach(), m.eps, B, H, W)
    _put(gd, m.weights, dw)
    dres = ops.gemm(dskip, _wb(_w2(m.pre_conv[0].weight).t()))
    _put(gd, m.pre_conv[0].weight, ops.gemm_tn(dskip, res, lazy=True))
    return dx, dres


@ops.keeps_precision
class WFFn(torch.autograd.Function):
    """WF (unetformer_mmsam.py:233-248): x fp32 [B*(H/2)*(W/2), C], res bf16 [B*H*W, Cin] (row order
    `perm`) -> ConvBNReLU6(fw0*pre_conv(res) + fw1*up2(x)) fp32 [B*H*W, C]."""

    @staticmethod
    def forward(ctx, x, res, m, perm, geom, training, *plist):
        x, res = x.contiguous(), res.contiguous()
        y, _, saved = _fuse_fwd(x, res, perm, m, geom, training)
        ctx.save_for_backward(x, res, *saved)
        ctx.m, ctx.perm, ctx.geom, ctx.training, ctx.plist = m, perm, geom, training, plist
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, *saved = ctx.saved_tensors
        gd = {}
        dx, dres = _fuse_bwd(dy.contiguous(), x, res, ctx.perm, ctx.m, ctx.geom, ctx.training, saved, gd)
        return (dx, dres, None, None, None, None) + _grads(ctx.plist, gd)


def frh_params(m):
    return wf_params(m) + [m.pa[0].weight, m.pa[0].bias, m.ca[1][0].weight, m.ca[3][0].weight, m.shortcut[0].weight,
                           m.shortcut[1].weight, m.shortcut[1].bias, m.proj[0].weight, m.proj[1].weight, m.proj[1].bias, m.proj[2].weight]


@ops.keeps_precision
class FRHFn(torch.autograd.Function):
    """FeatureRefinementHead (unetformer_mmsam.py:320-354) -> bf16 [B*H*W, C]."""

    @staticmethod
    def forward(ctx, x, res, m, perm, geom, training, *plist):
        B, H, W = geom
        P = H * W
        x, res = x.contiguous(), res.contiguous()
        xf, xfb, saved = _fuse_fwd(x, res, perm, m, geom, training, out2_dtype=BF)
        csc = ops.gemm(xfb, _wb(_w2(m.shortcut[0].weight)), out_dtype=F32)
        sc, _, msc, rsc = _bn(csc, m.shortcut[1], training)
        dwo = ops.dwconv_fwd(xf, m.pa[0].weight.detach().reshape(-1, 9), m.pa[0].bias, B, H, W, 3, False)
        pool, _ = ops.colsum(xf, B, P)
        z, g = ops.se_gate_fwd(pool, P, _w2(m.ca[1][0].weight), None, _w2(m.ca[3][0].weight), None, True)
        pc = ops.frh_gate_fwd(xf, dwo, g, B, P)
        dw2 = ops.dwconv_fwd(pc, m.proj[0].weight.detach().reshape(-1, 9), None, B, H, W, 3, False)
        pb, _, mp, rp = _bn(dw2, m.proj[1], training, out_dtype=BF)
        out = ops.gemm(pb, _wb(_w2(m.proj[2].weight)), residual=sc, act=5)
        ctx.save_for_backward(x, res, *saved, xf, xfb, csc, msc, rsc, dwo, pool, z, g, pc, dw2, mp, rp, pb, out)
        ctx.m, ctx.perm, ctx.geom, ctx.training, ctx.plist = m, perm, geom, training, plist
        return out

    @staticmethod
    def backward(ctx, dout):
        t = ctx.saved_tensors
        x, res, saved = t[0], t[1], t[2:7]
        xf, xfb, csc, msc, rsc, dwo, pool, z, g, pc, dw2, mp, rp, pb, out = t[7:]
        m, (B, H, W), training = ctx.m, ctx.geom, ctx.training
        P = H * W
        gd = {}
        dz = ops.act_bwd(dout.contiguous(), out, ACT_RELU6)                       # through the final ReLU6
        dpb = ops.gemm(dz, _wb(_w2(m.proj[2].weight).t()))
        _put(gd, m.proj[2].weight, ops.gemm_tn(dz, pb, lazy=True))
        ddw2, dg, dbt = _bn_bwd(dpb, dw2, mp, rp, m.proj[1], training)
        _put(gd, m.proj[1].weight, dg); _put(gd, m.proj[1].bias, dbt)
        dpc, dwp, _ = ops.dwconv_bwd(ddw2, pc, m.proj[0].weight.detach().reshape(-1, 9), B, H, W, 3, False)
        _put(gd, m.proj[0].weight, dwp)
        dgate = ops.coldot(dpc, xf, B, P)
        dpool, dW1, _, dW2, _ = ops.se_gate_bwd(dgate, pool, P, _w2(m.ca[1][0].weight), _w2(m.ca[3][0].weight), True, z, g, False)
        _put(gd, m.ca[1][0].weight, dW1); _put(gd, m.ca[3][0].weight, dW2)
        dxf, ddwo = ops.frh_gate_bwd(dpc, xf, dwo, g, dpool, B, P)
        dxf2, dwpa, dbpa = ops.dwconv_bwd(ddwo, xf, m.pa[0].weight.detach().reshape(-1, 9), B, H, W, 3, False, want_bias=True)
        _put(gd, m.pa[0].weight, dwpa); _put(gd, m.pa[0].bias, dbpa)
        dcsc, dg, dbt = _bn_bwd(dz, csc, msc, rsc, m.shortcut[1], training, dx_dtype=BF)
        _put(gd, m.shortcut[1].weight, dg); _put(gd, m.shortcut[1].bias, dbt)
        _put(gd, m.shortcut[0].weight, ops.gemm_tn(dcsc, xfb, lazy=True))
        dxf = ops.add3(dxf, dxf2)
        ops.gemm(dcsc, _wb(_w2(m.shortcut[0].weight).t()), residual=dxf, out=dxf)
        dx, dres = _fuse_bwd(dxf, x, res, ctx.perm, m, ctx.geom, training, saved, gd)
        return (dx, dres, None, None, None, None) + _grads(ctx.plist, gd)


# ---------------------------------------------------------------------------------------------- segmentation head
NCLS_PAD = 32      # class maps are computed 32 wide: GEMM N % 16 == 0 forward and K % 32 == 0 in the dgrad


def seghead_params(head):
    return [head[0][0].weight, head[0][1].weight, head[0][1].bias, head[2][0].weight]


@ops.keeps_precision
class SegHeadFn(torch.autograd.Function):
    """segmentation_head + final bilinear upsample (unetformer_mmsam.py:439-441,454-455): x bf16
    [B*H*W, C] -> logits fp32 [B, ncls, s*H, s*W]. drop_mask [B, C] of {0, 1/(1-p)} = Dropout2d."""

    @staticmethod
    def forward(ctx, x, head, geom, scale, drop_mask, training, *plist):
        B, H, W = geom
        x = x.contiguous()
        wcls = head[2][0].weight
        ncls = wcls.shape[0]
        if ncls > NCLS_PAD:
            raise NotImplementedError(f"num_classes > {NCLS_PAD}")
        c, cols = _conv3_fwd(x, head[0][0].weight, B, H, W)
        hb, _, mean, rstd = _bn(c, head[0][1], training, act=ACT_RELU6, out_dtype=BF, chan_scale=drop_mask, rpi=H * W)
        w2c = _w2(wcls)
        wp = derived(("cls", w2c.data_ptr()), (NCLS_PAD, wcls.shape[1]), BF, lambda t: [(t[:ncls], w2c)], device=x.device)
        lg = ops.gemm(hb, wp, out_dtype=F32)
        logits = ops.up_final_fwd(lg, B, H, W, ncls, scale)
        ctx.save_for_backward(cols, c, mean, rstd, hb, drop_mask)
        ctx.head, ctx.geom, ctx.scale, ctx.training, ctx.plist = head, geom, scale, training, plist
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        cols, c, mean, rstd, hb, drop_mask = ctx.saved_tensors
        head, (B, H, W), training = ctx.head, ctx.geom, ctx.training
        ncls = head[2][0].weight.shape[0]
        gd = {}
        dlg = ops.up_final_bwd(dlogits.contiguous(), NCLS_PAD, B, H, W, ncls, ctx.scale)
        w2c = _w2(head[2][0].weight)
        wpT = derived(("clsT", w2c.data_ptr()), (w2c.shape[1], NCLS_PAD), BF, lambda t: [(t[:, :ncls], w2c.t())], device=dlg.device)
        dhb = ops.gemm(dlg, wpT)
        _put(gd, head[2][0].weight, ops.gemm_tn(dlg, hb, lazy=True)[:ncls])
        dc, dg, dbt = _bn_bwd(dhb, c, mean, rstd, head[0][1], training, act=ACT_RELU6, dx_dtype=BF, chan_scale=drop_mask, rpi=H * W)
        _put(gd, head[0][1].weight, dg); _put(gd, head[0][1].bias, dbt)
        dx, dW3 = _conv3_bwd(dc, cols, head[0][0].weight, B, H, W)
        _put(gd, head[0][0].weight, dW3)
        return (ops.cast(dx, BF), None, None, None, None, None) + _grads(ctx.plist, gd)


# ---------------------------------------------------------------------------------------------- channels-last <-> NCHW glue
def to_rows(x_nchw, dtype=None):
    """(B,C,H,W) -> contiguous [B*H*W, C] (layout change only)."""
    B, C, H, W = x_nchw.shape
    t = x_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C)
    return t.contiguous() if dtype is None else t.to(ops._odt(dtype)).contiguous()


def to_nchw(rows, B, H, W):
    return rows.view(B, H, W, -1).permute(0, 3, 1, 2)


def layer_norm_nhwc(x, weight, bias, eps):
    """LayerNorm over the last dim of a channels-last tensor (forward only; LayerNorm2d interface)."""
    shp = x.shape
    y, _, _ = ops.layernorm_fwd(x.reshape(-1, shp[-1]).contiguous().float(), weight, bias, eps, out_dtype=F32, want_stats=False)
    return y.view(shp)
