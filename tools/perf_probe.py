#!/usr/bin/env python3
"""Per-kernel timing probe at the benchmark shape (GPU box): python tools/perf_probe.py [filter]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
OUT = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def probe_attn(B=2, rel_pos=False):
    HD = int(os.environ.get("PROBE_HD", "16"))             # head width (24: BASELINE configs[4], 32-wide slots)
    plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, HD, 0)
    Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
    qkvh = (torch.randn(Bw, h, 3, Lp, DP, device=dev) * 0.25).to(BF)
    qkvh[:, :, :, Lw:] = 0
    rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
    oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev)
    lse = torch.empty(Bw, h, Lp, device=dev)
    ls = torch.full((h,), 2.3, device=dev)
    bias = torch.randn(h, Lw, Lw, device=dev) if rel_pos else None
    doh = (torch.randn(Bw, h, Lp, DP, device=dev)).to(BF)
    dq = torch.empty(Bw, h, 3, Lp, DP, dtype=BF, device=dev)
    dls = torch.zeros(h, device=dev)
    dbias = torch.zeros(h, Lw, Lw, device=dev) if rel_pos else None
    pk = ops.attn_pack_bias(bias) if rel_pos else None
    for gen, dbg in (("first generation (csrc/attn.hip)", L.ATTN_FIRST_GEN), ("default", 0), ("statistics from LDS (backward)", L.ATTN_PLAIN_STATS)):
        a = ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, HD, plan.nwh, plan.nww, plan.mask_thr, max_chunks=32 if rel_pos else 64, bias_pack=pk)
        a.dbg = dbg
        say(f"attn_fwd  {gen} B={B} bias={rel_pos}: {timeit(lambda: ops.attn_fwd(a), n=20):.1f} us")
        a = ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, HD, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm,
                          dqkvh=dq, dlogit=dls, dbias=dbias, max_chunks=32 if rel_pos else 256 // h, bias_pack=pk)     # (as swv2_block_bwd)
        a.dbg = dbg
        say(f"attn_bwd  {gen} B={B} bias={rel_pos}: {timeit(lambda: ops.attn_bwd(a), n=20):.1f} us")


def probe_gemm(B=2):
    T, Cc = 64800, 128
    M = B * T
    x = torch.randn(M, Cc, device=dev)
    xb = x.to(BF)
    for N in (128, 384, 512):
        w = ops.prep_weight(torch.randn(N, Cc, device=dev))
        o = torch.empty(M, N, dtype=BF, device=dev)
        t = timeit(lambda: ops.linear(ops.op_f32(x), w, ops.epilogue(L.EPI_BF16, o, ld=N), N))
        say(f"linear f32[{M},{Cc}] -> bf16 N={N}: {t:.1f} us  ({(M * Cc * 4 + M * N * 2) / t / 1e3:.0f} GB/s)")
        t = timeit(lambda: ops.linear(ops.op_bf16(xb), w, ops.epilogue(L.EPI_BF16, o, ld=N), N))
        say(f"linear bf16[{M},{Cc}] -> bf16 N={N}: {t:.1f} us  ({(M * Cc * 2 + M * N * 2) / t / 1e3:.0f} GB/s)")
    hb = torch.randn(M, 512, device=dev).to(BF)
    w2 = ops.prep_weight(torch.randn(Cc, 512, device=dev))
    o = torch.empty(M, Cc, dtype=BF, device=dev)
    t = timeit(lambda: ops.linear(ops.op_bf16(hb), w2, ops.epilogue(L.EPI_BF16, o, ld=Cc), Cc))
    say(f"linear bf16[{M},512] -> bf16 N=128: {t:.1f} us ({(M * 512 * 2 + M * Cc * 2) / t / 1e3:.0f} GB/s)")
    t = timeit(lambda: ops.linear(ops.op_bf16(hb, gelu=True), w2, ops.epilogue(L.EPI_BF16, o, ld=Cc), Cc))
    say(f"linear gelu(bf16[{M},512]) -> bf16 N=128: {t:.1f} us")
    dW = torch.zeros(512, Cc, device=dev)
    db = torch.zeros(512, device=dev)
    for splits in (64, 128, 256, 512):
      for wsp in (False, True):
        t = timeit(lambda: ops.linear_wgrad(ops.op_bf16(hb), ops.op_f32(x), dW, db, splits=splits, workspace=wsp))
        say(f"wgrad dY bf16[{M},512] x X f32[{M},128] splits={splits} ws={wsp}: {t:.1f} us ({M * (1024 + 512) / t / 1e3:.0f} GB/s)")
    da2 = torch.randn(M, Cc, device=dev).to(BF)
    dW2 = torch.zeros(Cc, 512, device=dev)
    for splits in (64, 128, 256, 512):
      for wsp in (False, True):
        t = timeit(lambda: ops.linear_wgrad(ops.op_bf16(da2), ops.op_bf16(hb), dW2, None, splits=splits, workspace=wsp))
        say(f"wgrad dY bf16[{M},128] x X bf16[{M},512] splits={splits} ws={wsp}: {t:.1f} us ({M * (1024 + 256) / t / 1e3:.0f} GB/s)")
    for splits in (64, 128, 256, 512):
      for wsp in (False, True):
        dWp = torch.zeros(Cc, Cc, device=dev)
        t = timeit(lambda: ops.linear_wgrad(ops.op_bf16(da2), ops.op_bf16(xb), dWp, None, splits=splits, workspace=wsp))
        say(f"wgrad dY bf16[{M},128] x X bf16[{M},128] splits={splits} ws={wsp}: {t:.1f} us ({M * 512 / t / 1e3:.0f} GB/s)")
    a = xb
    y = torch.empty(M, Cc, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    g, bt = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    t = timeit(lambda: ops.ln_residual_fwd(a, x, g, bt, None, None, y, mean, rstd, M, Cc, 0, T))
    say(f"ln_residual_fwd: {t:.1f} us ({M * Cc * (2 + 4 + 4) / t / 1e3:.0f} GB/s)")
    da = torch.empty(M, Cc, dtype=BF, device=dev)
    dg, dbt = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    t = timeit(lambda: ops.ln_residual_bwd(a, x, g, None, None, mean, rstd, da, dg, dbt, M, Cc, T))
    say(f"ln_residual_bwd: {t:.1f} us ({M * Cc * (2 + 4 + 2) / t / 1e3:.0f} GB/s)")


def probe_mlp(B=2, hid=512):
    import ctypes
    T, Cc = 64800, 128
    M = B * T
    x = torch.randn(M, Cc, device=dev)
    w1, w2 = ops.prep_weight(torch.randn(hid, Cc, device=dev) * 0.1), ops.prep_weight(torch.randn(Cc, hid, device=dev) * 0.1)
    b1, b2, g, bt = torch.zeros(hid, device=dev), torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    y, hpre, a2 = torch.empty(M, Cc, device=dev), torch.empty(M, hid, dtype=BF, device=dev), torch.empty(M, Cc, dtype=BF, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    a = L.MlpArgs()
    a.x, a.w1, a.b1, a.w2, a.b2, a.gamma, a.beta = (t.data_ptr() for t in (x, w1, b1, w2, b2, g, bt))
    a.hpre, a.a2, a.mean, a.rstd, a.y = (t.data_ptr() for t in (hpre, a2, mean, rstd, y))
    a.M, a.C, a.hidden, a.rows_per_sample, a.eps = M, Cc, hid, T, 1e-5
    fn, st = L.load().swv2_mlp_fwd, torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: fn(ctypes.byref(a), st), n=20)
    say(f"mlp_fwd M={M} hidden={hid}: {t:.1f} us ({M * (Cc * 4 * 2 + hid * 2 + Cc * 2) / t / 1e3:.0f} GB/s)")
    dy, da2, dh, dx = torch.randn(M, Cc, device=dev), torch.empty(M, Cc, dtype=BF, device=dev), torch.empty(M, hid, dtype=BF, device=dev), torch.empty(M, Cc, device=dev)
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    ws = torch.empty(L.load().swv2_mlp_bwd_ws_floats(M, Cc), device=dev)
    w2t, w1t = ops.prep_weight(torch.randn(Cc, hid, device=dev) * 0.1, transpose=True), ops.prep_weight(torch.randn(hid, Cc, device=dev) * 0.1, transpose=True)
    fn(ctypes.byref(a), st)
    b = L.MlpBwdArgs()
    b.dy, b.a2, b.mean, b.rstd, b.gamma, b.hpre, b.w2t, b.w1t = (t_.data_ptr() for t_ in (dy, a2, mean, rstd, g, hpre, w2t, w1t))
    b.da2, b.dh, b.dx, b.dgamma, b.dbeta, b.ws = (t_.data_ptr() for t_ in (da2, dh, dx, dg, db, ws))
    b.M, b.C, b.hidden, b.rows_per_sample = M, Cc, hid, T
    fb = L.load().swv2_mlp_bwd
    t = timeit(lambda: fb(ctypes.byref(b), st), n=20)
    say(f"mlp_bwd M={M} hidden={hid}: {t:.1f} us ({M * (Cc * 4 * 2 + hid * 2 * 2 + Cc * 2 * 2) / t / 1e3:.0f} GB/s)")


if __name__ == "__main__":
    if "mlp" in (sys.argv[1] if len(sys.argv) > 1 else ""):
        for hid in ([128, 256, 512, 1024] if "sweep" in sys.argv[1] else [512]):      # (mlp_sweep: cost per 32-unit weight chunk vs fixed cost)
            probe_mlp(2, hid)
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    if "attn" in flt or not flt:
        probe_attn(2, False)
        probe_attn(2, True)
        probe_attn(8, False)
    if "gemm" in flt or not flt:
        probe_gemm(2)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "perf_probe.txt"), "w").write("\n".join(OUT) + "\n")
