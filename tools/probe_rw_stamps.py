#!/usr/bin/env python3
"""In-kernel phase timing of the resident-weight GEMM (gemm_rw_kernel): private build with -DSWV2_RW_STAMPS; wave 0 of
every workgroup sums s_memtime deltas per phase -- GPU box, diagnostics only."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = "/tmp/libswv2_rwstamps.so"
srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_RW_STAMPS", "-o", so] + srcs)
L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = 2
plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, 16, 0)
Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
Mw, Cc = Bw * Lp, 128
x = torch.randn(B * 64800, Cc, device=dev)
w = ops.prep_weight(torch.randn(3 * Cc, Cc, device=dev) * 0.1, row_map=plan.qkv_map, out_rows=3 * h * DP)
qkvh = torch.empty(Bw, h, 3, Lp, DP, dtype=BF, device=dev)
rnorm = torch.empty(Bw, h, 2, Lp, device=dev)
bias = torch.zeros(3 * h * DP, device=dev)
names = ["wait for A loads", "commit (cvt + LDS write)", "barrier 1", "issue next + resolve", "LDS reads + MFMA", "barrier 2", "epilogue", "barrier 3"]
lib = ctypes.CDLL(so)


def run(tag, fn, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 100
    buf = torch.zeros(256 * 8, dtype=torch.int64)
    assert lib.swv2_debug_rw_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    st = buf.view(256, 8).double()
    tot = st.sum(1)
    print(f"{tag}: {t:.1f} us ({nbytes / t / 1e3:.0f} GB/s); wave 0 of 256 workgroups, total ticks mean {tot.mean():.0f}")
    for i, n in enumerate(names):
        print(f"  {n:28s} {100 * st[:, i].mean() / tot.mean():5.1f} %  (min {st[:, i].min():.0f} max {st[:, i].max():.0f})")


e = ops.epilogue(L.EPI_QKV_HEADS, qkvh, bias=bias, aux_out=rnorm, p=(h, 0, Lp, DP, Lw))
a = ops.op_f32(x, rows=Mw, rowidx=plan.rowidx)
run("qkv forward (gathered f32 rows -> head layout)", lambda: ops.linear(a, w, e, 3 * h * DP), Mw * Cc * 4 + Mw * 3 * Cc * 2)
dq = torch.randn(Bw, h, 3, Lp, DP, device=dev).to(BF)
wt = ops.prep_weight(torch.randn(3 * Cc, Cc, device=dev) * 0.1, transpose=True, col_map=plan.qkv_map, out_cols=3 * h * DP)
dx1 = torch.randn(B * 64800, Cc, device=dev)
dx = torch.empty_like(dx1)
e2 = ops.epilogue(L.EPI_F32, dx, ld=Cc, aux=dx1, rowidx=plan.rowidx)
a2 = ops.op_heads(dq, Bw, h, 3, Lp, DP)
run("d(qkv) -> dx (head layout -> scattered f32 rows + residual)", lambda: ops.linear(a2, wt, e2, Cc), Mw * 3 * Cc * 2 + 2 * Mw * Cc * 4)
