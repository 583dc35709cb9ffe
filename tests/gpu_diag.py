#!/usr/bin/env python3
"""GPU diagnostic sweep: runs every kernel family against the CPU oracle and prints error metrics without stopping at
the first failure.  Usage (on the GPU box):  python tests/gpu_diag.py [filter]  -> also writes gpurun_out/diag.txt
"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import swin_oracle as O  # noqa: E402
from swin_v2_weather_amd import _lib as L, ops  # noqa: E402
from swin_v2_weather_amd.networks import swinv2_global as N  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")
BF = torch.bfloat16
LINES = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LINES.append(s)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def mx(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def rb(t):
    """round to bf16 and back (what the kernels see)"""
    return t.to(BF).float()


def to_heads(x, Bw, Lw, h, d, Lp, DP, parts):
    """[Bw, L, parts*h*d] -> [Bw][h][parts][Lp][DP] zero padded (float)"""
    x = x.reshape(Bw, Lw, parts, h, d).permute(0, 3, 2, 1, 4)
    out = torch.zeros(Bw, h, parts, Lp, DP, dtype=x.dtype)
    out[:, :, :, :Lw, :d] = x
    return out


def from_heads(xh, Bw, Lw, h, d, parts):
    """inverse of to_heads -> [Bw, L, parts*h*d]"""
    return xh[:, :, :, :Lw, :d].permute(0, 3, 2, 1, 4).reshape(Bw, Lw, parts * h * d)


# ---------------------------------------------------------------------------------------------
def t_attn(wh, ww, h, d, B, nwh, nww, shifted, use_bias, seed=0):
    torch.manual_seed(seed)
    Lw = wh * ww
    Lp, DP = ops.attn_geometry(Lw, d)
    nW = nwh * nww
    Bw = B * nW
    Cc = h * d
    qkv = torch.randn(Bw, Lw, 3 * Cc)
    ls = torch.log(torch.tensor(10.0)) + 0.5 * torch.randn(h)
    ls[-1] = 5.0
    bias = torch.randn(h, Lw, Lw) if use_bias else None
    sh = wh // 2 if shifted else 0
    gh, gw = nwh * wh, nww * ww
    mask = O.shift_mask(gh, gw, wh, ww, sh, ww // 2 if shifted else 0)
    mask_thr = (wh - sh) * ww if sh > 0 else 0
    # the kernel consumes normalised q, k (bf16) + v (bf16); build them like the QKV epilogue does
    q, k, v = qkv.reshape(Bw, Lw, 3, Cc).unbind(2)
    qh = q.reshape(Bw, Lw, h, d)
    kh = k.reshape(Bw, Lw, h, d)
    rq = 1.0 / qh.norm(dim=-1).clamp_min(1e-12)
    rk = 1.0 / kh.norm(dim=-1).clamp_min(1e-12)
    qn = rb(qh * rq.unsqueeze(-1)).reshape(Bw, Lw, Cc)
    kn = rb(kh * rk.unsqueeze(-1)).reshape(Bw, Lw, Cc)
    vb = rb(v)
    qkvh = to_heads(torch.stack([qn, kn, vb], 2).reshape(Bw, Lw, 3 * Cc), Bw, Lw, h, d, Lp, DP, 3).to(BF).to(dev).contiguous()
    rnorm = torch.zeros(Bw, h, 2, Lp)
    rnorm[:, :, 0, :Lw] = rq.permute(0, 2, 1)
    rnorm[:, :, 1, :Lw] = rk.permute(0, 2, 1)
    oh = torch.full((Bw, h, Lp, DP), float("nan"), dtype=BF, device=dev)
    lse = torch.zeros(Bw, h, Lp, device=dev)
    lsd = ls.to(dev)
    bd = bias.to(dev).contiguous() if use_bias else None
    args = ops.attn_args(qkvh, lsd, bd, oh, lse, Bw, h, Lw, d, nwh, nww, mask_thr)
    ops.attn_fwd(args)
    torch.cuda.synchronize()
    # reference on the SAME (already normalised, bf16-rounded) operands, taken as free variables (no re-normalisation):
    # S = sigma * qn kn^T + bias + mask ; softmax ; P v   (oracle/swin_oracle.py attention_core without its normalise)
    qkv_ref = torch.stack([qn, kn, vb], 2).reshape(Bw, Lw, 3 * Cc).double().requires_grad_(True)
    ls_ref = ls.double().requires_grad_(True)
    bias_ref = bias.double().requires_grad_(True) if use_bias else None
    q_, k_, v_ = qkv_ref.reshape(Bw, Lw, 3, h, d).permute(2, 0, 3, 1, 4)
    S = torch.einsum("bhqd,bhkd->bhqk", q_, k_) * torch.exp(torch.clamp(ls_ref, max=O.LOGIT_MAX)).view(1, h, 1, 1)
    if use_bias:
        S = S + bias_ref.unsqueeze(0)
    if mask is not None:
        S = (S.reshape(B, nW, h, Lw, Lw) + mask.double().view(1, nW, 1, Lw, Lw)).reshape(Bw, h, Lw, Lw)
    o_ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(S, -1), v_).reshape(Bw, Lw, Cc)
    o_gpu = from_heads(oh.float().cpu().unsqueeze(2), Bw, Lw, h, d, 1)
    tag = f"attn L={Lw} h={h} d={d} Bw={Bw} shift={shifted} bias={use_bias}"
    say(f"{tag}: fwd rel {rel(o_gpu, o_ref):.3e} max {mx(o_gpu, o_ref):.3e} nan={bool(torch.isnan(o_gpu).any())}")
    pad = oh.float().cpu()
    say(f"   padded rows zero: {float(pad[:, :, Lw:, :].abs().max()) if Lp > Lw else 0.0} padded cols zero: "
        f"{float(pad[:, :, :, d:].abs().max()) if DP > d else 0.0}")
    # backward.  The oracle differentiates w.r.t. the normalised-operand tensor; the kernel returns the gradient
    # w.r.t. the UN-normalised q,k, so compare dv directly and dq,dk through the normalisation Jacobian.
    go = torch.randn(Bw, Lw, Cc)
    gob = rb(go)
    o_ref.backward(gob.double())
    doh = to_heads(gob, Bw, Lw, h, d, Lp, DP, 1).squeeze(2).to(BF).to(dev).contiguous()
    dqkvh = torch.full((Bw, h, 3, Lp, DP), float("nan"), dtype=BF, device=dev)
    dls = torch.zeros(h, device=dev)
    dbias = torch.zeros(h, Lw, Lw, device=dev) if use_bias else None
    args = ops.attn_args(qkvh, lsd, bd, oh, lse, Bw, h, Lw, d, nwh, nww, mask_thr, doh=doh, rnorm=rnorm.to(dev).contiguous(),
                         dqkvh=dqkvh, dlogit=dls, dbias=dbias)
    ops.attn_bwd(args)
    torch.cuda.synchronize()
    dq_gpu = from_heads(dqkvh.float().cpu(), Bw, Lw, h, d, 3).reshape(Bw, Lw, 3, Cc)
    g = qkv_ref.grad.reshape(Bw, Lw, 3, Cc)          # grads w.r.t. (qn, kn, v) as free variables incl. re-normalisation

    def through_norm(gn, xn, r):                      # d/dx of x/|x| applied to gn, expressed with xn = x/|x|, r = 1/|x|
        gn = gn.reshape(Bw, Lw, h, d)
        xn = xn.reshape(Bw, Lw, h, d).double()
        return (r.unsqueeze(-1).double() * (gn - xn * (gn * xn).sum(-1, keepdim=True))).reshape(Bw, Lw, Cc)
    dq_ref = through_norm(g[:, :, 0], qn, rq)
    dk_ref = through_norm(g[:, :, 1], kn, rk)
    say(f"   bwd dq rel {rel(dq_gpu[:, :, 0], dq_ref):.3e}  dk rel {rel(dq_gpu[:, :, 1], dk_ref):.3e}  "
        f"dv rel {rel(dq_gpu[:, :, 2], g[:, :, 2]):.3e}  dlogit rel {rel(dls, ls_ref.grad):.3e} "
        f"({dls.cpu().numpy()[:3]} vs {ls_ref.grad.numpy()[:3]})")
    if use_bias:
        say(f"   dbias rel {rel(dbias, bias_ref.grad):.3e} max {mx(dbias, bias_ref.grad):.3e}")


def t_linear():
    torch.manual_seed(1)
    M, K, Nn = 300, 128, 200
    x = torch.randn(M, K)
    w = torch.randn(Nn, K) * 0.1
    b = torch.randn(Nn)
    wb = ops.prep_weight(w.to(dev))
    say(f"prep_weight cast: max {mx(wb.float(), rb(w)):.3e}")
    ref = rb(x) @ rb(w).T + b
    out = torch.full((M, Nn), float("nan"), dtype=BF, device=dev)
    ops.linear(ops.op_f32(x.to(dev)), wb, ops.epilogue(L.EPI_BF16, out, ld=Nn, bias=b.to(dev)), Nn)
    say(f"linear f32->bf16 M={M} K={K} N={Nn}: rel {rel(out.float(), ref):.3e} nan={bool(torch.isnan(out.float()).any())}")
    out32 = torch.full((M, Nn), float("nan"), device=dev)
    ops.linear(ops.op_bf16(x.to(BF).to(dev)), wb, ops.epilogue(L.EPI_F32, out32, ld=Nn, bias=b.to(dev)), Nn)
    say(f"linear bf16->f32: rel {rel(out32, ref):.3e}")
    # larger K with a tail, aux add, row gather / scatter
    M, K, Nn = 1000, 1168, 96
    x = torch.randn(M, K)
    w = torch.randn(Nn, K) * 0.05
    ref = rb(x) @ rb(w).T
    idx = torch.randperm(M)
    rowidx = idx.to(torch.int32).to(dev)
    aux = torch.randn(M, Nn)
    out32 = torch.full((M, Nn), float("nan"), device=dev)
    ops.linear(ops.op_f32(x.to(dev), rowidx=rowidx), ops.prep_weight(w.to(dev)),
               ops.epilogue(L.EPI_F32, out32, ld=Nn, aux=aux.to(dev)), Nn)
    say(f"linear gather K=1168 N=96 +aux: rel {rel(out32, ref[idx] + aux):.3e}")
    out32 = torch.full((M, Nn), float("nan"), device=dev)
    ops.linear(ops.op_f32(x.to(dev)), ops.prep_weight(w.to(dev)), ops.epilogue(L.EPI_F32, out32, ld=Nn, rowidx=rowidx), Nn)
    exp = torch.empty(M, Nn)
    exp[idx] = ref
    say(f"linear scatter: rel {rel(out32, exp):.3e}")
    # gelu on load + gelu grad epilogue
    M, K, Nn = 500, 512, 128
    hpre = torch.randn(M, K)
    w = torch.randn(Nn, K) * 0.05
    out = torch.empty(M, Nn, dtype=BF, device=dev)
    ops.linear(ops.op_bf16(hpre.to(BF).to(dev), gelu=True), ops.prep_weight(w.to(dev)), ops.epilogue(L.EPI_BF16, out, ld=Nn), Nn)
    say(f"linear gelu-on-load: rel {rel(out.float(), rb(O.gelu_erf(rb(hpre))) @ rb(w).T):.3e}")
    dy = torch.randn(M, Nn)
    dh = torch.empty(M, K, dtype=BF, device=dev)
    ops.linear(ops.op_bf16(dy.to(BF).to(dev)), ops.prep_weight(w.to(dev), transpose=True),
               ops.epilogue(L.EPI_GELU_GRAD, dh, ld=K, aux=hpre.to(BF).to(dev)), K)
    hp = rb(hpre).double().requires_grad_(True)
    O.gelu_erf(hp).backward((rb(dy) @ rb(w)).double())
    say(f"linear gelu-grad epilogue: rel {rel(dh.float(), hp.grad):.3e}")
    # weight gradient
    dW = torch.zeros(Nn, K, device=dev)
    db = torch.zeros(Nn, device=dev)
    ops.linear_wgrad(ops.op_bf16(dy.to(BF).to(dev)), ops.op_bf16(hpre.to(BF).to(dev)), dW, db, splits=7, workspace=False)
    say(f"wgrad bf16xbf16 M={M}: dW rel {rel(dW, rb(dy).T @ rb(hpre)):.3e} db rel {rel(db, rb(dy).sum(0)):.3e}")
    dW = torch.zeros(Nn, K, device=dev)
    ops.linear_wgrad(ops.op_bf16(dy.to(BF).to(dev)), ops.op_bf16(hpre.to(BF).to(dev), gelu=True), dW, None, splits=3)
    say(f"wgrad gelu-on-load: dW rel {rel(dW, rb(dy).T @ rb(O.gelu_erf(rb(hpre)))):.3e}")


def t_patch():
    torch.manual_seed(2)
    B, Cin, H, W, Cc = 2, 7, 24, 40, 32
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cc, Cin, 4, 4) * 0.1
    b = torch.randn(Cc)
    out = torch.empty(B * 6 * 10, Cc, dtype=BF, device=dev)
    ops.linear(ops.op_patch(x.to(dev)), ops.prep_weight(w.to(dev)), ops.epilogue(L.EPI_BF16, out, ld=Cc, bias=b.to(dev)), Cc)
    ref = torch.nn.functional.conv2d(rb(x), rb(w), b, stride=4).permute(0, 2, 3, 1).reshape(-1, Cc)
    say(f"patch im2col GEMM: rel {rel(out.float(), ref):.3e}")
    # un-patchify epilogue with skip
    Cout = 5
    e = torch.randn(B * 60, Cc)
    wh_ = torch.randn(Cout * 16, Cc) * 0.1
    perm = (torch.arange(16).view(1, -1) * Cout + torch.arange(Cout).view(-1, 1)).reshape(-1).to(torch.int32).to(dev)
    y = torch.full((B, Cout, H, W), float("nan"), device=dev)
    ops.linear(ops.op_f32(e.to(dev)), ops.prep_weight(wh_.to(dev), row_map=perm),
               ops.epilogue(L.EPI_UNPATCH, y, aux=x.to(dev), p=(Cout, H, W, Cin, 0)), Cout * 16)
    ref = O.head_unpatchify(rb(e).reshape(B, 6, 10, Cc), rb(wh_), 4, Cout, x)
    say(f"head + un-patchify + skip: rel {rel(y, ref):.3e} nan={bool(torch.isnan(y).any())}")


def t_ln():
    torch.manual_seed(3)
    for Cc in (32, 96, 128, 192, 768):
        M = 777
        a = torch.randn(M, Cc) * 2 + 0.5
        res = torch.randn(M, Cc)
        g, bt = torch.randn(Cc), torch.randn(Cc)
        B = 3
        scale = torch.tensor([0.0, 1.25, 1.25])
        y = torch.full((M, Cc), float("nan"), device=dev)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        ab = a.to(BF).to(dev)
        ops.ln_residual_fwd(ab, res.to(dev), g.to(dev), bt.to(dev), scale.to(dev), None, y, mean, rstd, M, Cc, 0, M // B)
        ad = rb(a).double().requires_grad_(True)
        gd, bd = g.double().requires_grad_(True), bt.double().requires_grad_(True)
        sc = scale[(torch.arange(M) // (M // B)).clamp(max=B - 1)].double().view(-1, 1)
        ref = res.double() + sc * O.layer_norm(ad, gd, bd)
        dy = torch.randn(M, Cc)
        ref.backward(dy.double())
        da = torch.empty(M, Cc, dtype=BF, device=dev)
        dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        ops.ln_residual_bwd(ab, dy.to(dev), g.to(dev), scale.to(dev), None, mean, rstd, da, dg, db, M, Cc, M // B)
        say(f"ln_residual C={Cc}: fwd rel {rel(y, ref):.3e}  da rel {rel(da.float(), ad.grad):.3e} dgamma rel "
            f"{rel(dg, gd.grad):.3e} dbeta rel {rel(db, bd.grad):.3e}")


def load_into(module, fx, prefix="p:"):
    sd = {k[len(prefix):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=True), None
    return sd


def cmp_grads(module, fx, what):
    worst = 0.0
    for n, p in module.named_parameters():
        key = "g:" + n
        if key not in fx.files:
            continue
        r = rel(p.grad, torch.from_numpy(fx[key])) if p.grad is not None else float("nan")
        worst = max(worst, r) if r == r else float("nan")
        if r != r or r > 3e-2:
            say(f"      grad {n}: rel {r:.3e} |ref| {float(np.abs(fx[key]).max()):.3e}")
    say(f"   {what}: worst param-grad rel {worst:.3e}")


def emu_block(fx, tag, relpos, dims):
    """oracle forward/backward of a block fixture with bf16 rounding emulated at the HIP path's storage points"""
    gh, gw, wh, ww, sh, sw, Cc, h = dims
    cfg = O.SwinCfg(img_size=(gh * 4, gw * 4), patch_size=4, depth=2, num_heads=h, in_chans=1, out_chans=1, embed_dim=Cc,
                    window_ratio=1, rel_pos=relpos)

    class _C(O.SwinCfg):
        window = property(lambda s_: (wh, ww))

        def shift(s_, i):
            return (sh, sw)

        def drop_path(s_, i):
            return 0.0
    cfg.__class__ = _C
    p = {"b." + k[2:]: torch.from_numpy(fx[k]).clone().requires_grad_(True) for k in fx.files if k.startswith("p:")}
    x = torch.from_numpy(fx["x"]).clone().requires_grad_(True)
    O.set_rounding(O.bf16_round)
    try:
        y = O.block_forward(x, p, "b.", cfg, 1, training=False)
        y.backward(torch.from_numpy(fx["gy"]))
    finally:
        O.set_rounding(None)
    return y.detach(), x.grad, {k[2:]: v.grad for k, v in p.items()}


def t_block():
    for tag in ["nopos_noshift_eval", "relpos_shift_eval", "nopos_shift_3x3_eval"]:
        fx = np.load(os.path.join(GOLD, f"block_{tag}.npz"))
        gh, gw, wh, ww, sh, sw, Cc, h, B, seed, rng_seed, train = [int(v) for v in fx["meta"]]
        blk = N.SwinTransformerV2CrBlock(dim=Cc, num_heads=h, feat_size=(gh, gw), window_size=(wh, ww), shift_size=(sh, sw),
                                         rel_pos="relpos" in tag, drop_path=0.0)
        load_into(blk, fx)
        blk = blk.to(dev).eval()
        x = torch.from_numpy(fx["x"]).to(dev).requires_grad_(True)
        y = blk(x)
        y.backward(torch.from_numpy(fx["gy"]).to(dev))
        say(f"block[{tag}] vs fp32 reference fixture: y rel {rel(y, torch.from_numpy(fx['y'])):.3e} max "
            f"{mx(y, torch.from_numpy(fx['y'])):.3e}  gx rel {rel(x.grad, torch.from_numpy(fx['gx'])):.3e}")
        cmp_grads(blk, fx, f"block[{tag}] vs fixture")
        ye, gxe, ge = emu_block(fx, tag, "relpos" in tag, (gh, gw, wh, ww, sh, sw, Cc, h))
        say(f"block[{tag}] vs bf16-emulating oracle: y rel {rel(y, ye):.3e} max {mx(y, ye):.3e} gx rel {rel(x.grad, gxe):.3e}")
        worst = 0.0
        for n, p_ in blk.named_parameters():
            if ge.get(n) is None or p_.grad is None:
                continue
            r = rel(p_.grad, ge[n])
            if float(ge[n].abs().max()) > 1e-3:
                worst = max(worst, r)
            if r > 2e-2:
                say(f"      emu grad {n}: rel {r:.3e} |ref| {float(ge[n].abs().max()):.3e}")
        say(f"   block[{tag}] vs emu: worst param-grad rel {worst:.3e}")


def t_patch_modules():
    fx = np.load(os.path.join(GOLD, "patch_embed.npz"))
    pe = N.PatchEmbed(img_size=(24, 40), patch_size=4, in_chans=7, embed_dim=32, norm_layer=torch.nn.LayerNorm)
    load_into(pe, fx)
    pe = pe.to(dev)
    x = torch.from_numpy(fx["x"]).to(dev).requires_grad_(True)
    y = pe(x)
    y.backward(torch.from_numpy(fx["gy"]).to(dev))
    say(f"PatchEmbed: y rel {rel(y, torch.from_numpy(fx['y'])):.3e} gx rel {rel(x.grad, torch.from_numpy(fx['gx'])):.3e}")
    cmp_grads(pe, fx, "PatchEmbed")
    fx = np.load(os.path.join(GOLD, "patch_merging.npz"))
    pm = N.PatchMerging(dim=16)
    load_into(pm, fx)
    pm = pm.to(dev)
    x = torch.from_numpy(fx["x"]).to(dev).requires_grad_(True)
    y = pm(x)
    y.backward(torch.from_numpy(fx["gy"]).to(dev))
    say(f"PatchMerging: y rel {rel(y, torch.from_numpy(fx['y'])):.3e} gx rel {rel(x.grad, torch.from_numpy(fx['gx'])):.3e}")
    cmp_grads(pm, fx, "PatchMerging")


def t_model():
    for tag in ["nopos", "relpos_residual"]:
        fx = np.load(os.path.join(GOLD, f"model_{tag}.npz"))
        cin, cout, H, W, Cc, depth, h, ratio, relpos, residual, seed = [int(v) for v in fx["meta"]]
        m = N.SwinTransformerV2Cr(img_size=(H, W), patch_size=4, depths=(depth,), num_heads=(h,), in_chans=cin,
                                  out_chans=cout, embed_dim=Cc, img_window_ratio=ratio, full_pos_embed=True,
                                  rel_pos=bool(relpos), residual=bool(residual))
        load_into(m, fx)
        m = m.to(dev).eval()
        x = torch.from_numpy(fx["x"]).to(dev).requires_grad_(True)
        y = m(x)
        y.backward(torch.from_numpy(fx["gy"]).to(dev))
        say(f"model[{tag}] vs fp32 fixture: y rel {rel(y, torch.from_numpy(fx['y'])):.3e} max {mx(y, torch.from_numpy(fx['y'])):.3e} "
            f"gx rel {rel(x.grad, torch.from_numpy(fx['gx'])) if x.grad is not None else 'none'}")
        # bf16-emulating oracle
        cfg = O.SwinCfg(img_size=(H, W), patch_size=4, depth=depth, num_heads=h, in_chans=cin, out_chans=cout, embed_dim=Cc,
                        window_ratio=ratio, rel_pos=bool(relpos), residual=bool(residual))
        p = {k[2:]: torch.from_numpy(fx[k]).clone().requires_grad_(True) for k in fx.files if k.startswith("p:")}
        xo = torch.from_numpy(fx["x"]).clone().requires_grad_(True)
        O.set_rounding(O.bf16_round)
        try:
            yo = O.model_forward(xo, p, cfg, training=False)
            yo.backward(torch.from_numpy(fx["gy"]))
        finally:
            O.set_rounding(None)
        say(f"model[{tag}] vs bf16-emulating oracle: y rel {rel(y, yo):.3e} max {mx(y, yo):.3e} gx rel {rel(x.grad, xo.grad):.3e}")
        worst = 0.0
        for n, p_ in m.named_parameters():
            if p[n].grad is None or p_.grad is None:
                continue
            r = rel(p_.grad, p[n].grad)
            if float(p[n].grad.abs().max()) > 1e-3:
                worst = max(worst, r)
            if r > 3e-2:
                say(f"      emu grad {n}: rel {r:.3e} |ref| {float(p[n].grad.abs().max()):.3e}")
        say(f"   model[{tag}] vs emu: worst param-grad rel {worst:.3e}")


TESTS = [
    ("linear", t_linear),
    ("ln", t_ln),
    ("patch", t_patch),
    ("attn_small", lambda: [t_attn(6, 9, 4, 12, 2, 2, 2, s, b) for s in (False, True) for b in (False, True)]),
    ("attn_d32", lambda: t_attn(6, 9, 3, 32, 1, 2, 2, True, True)),
    ("attn_big", lambda: [t_attn(9, 18, 8, 16, 1, 2, 3, s, b) for s, b in ((False, False), (True, True))]),
    ("attn_d24", lambda: t_attn(9, 18, 2, 24, 1, 2, 2, True, False)),
    ("block", t_block),
    ("patch_modules", t_patch_modules),
    ("model", t_model),
]

if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    say("device:", torch.cuda.get_device_name(0), "lib version", L.load().swv2_version())
    for name, fn in TESTS:
        if flt and flt not in name:
            continue
        say(f"=== {name}")
        t0 = time.time()
        try:
            fn()
            torch.cuda.synchronize()
        except Exception:
            say("EXCEPTION in", name)
            say(traceback.format_exc())
        say(f"    ({time.time() - t0:.1f}s)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag.txt"), "w") as f:
        f.write("\n".join(LINES) + "\n")
