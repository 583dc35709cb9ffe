#!/usr/bin/env python3
"""In-kernel phase timing of the weight-gradient (TN) kernels: builds a private copy of the library with -DSWV2_TN_STAMPS
(wave 0 of every workgroup leaves its per-phase s_memtime sums at the head of its partial tile) -- GPU box, diagnostics."""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
so = "/tmp/libswv2_tnstamps.so"
srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_TN_STAMPS", "-o", so] + srcs)
L.LIB_PATH = so
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M, Cc, hid = B * 64800, 128, 512
names = ["wait for loads", "commit (cvt + LDS write)", "barrier 1", "issue next", "LDS reads + MFMA", "barrier 2", "partial tile store", "prologue"]


def report(tag, ws, ntiles_slices, t_us):
    st = ws.view(torch.int64).view(-1, 128 * 128 // 2)[:ntiles_slices, :8].cpu().double()
    tot = st.sum(1)
    print(f"{tag}: {t_us:.1f} us; wave 0 of {len(st)} workgroups, total ticks mean {tot.mean():.0f} (100 MHz s_memtime: {tot.mean() / 100:.1f} us)")
    for i, n in enumerate(names):
        print(f"  {n:28s} {st[:, i].mean() / 100:8.2f} us {100 * st[:, i].mean() / tot.mean():5.1f} %  (min {st[:, i].min() / 100:.2f} max {st[:, i].max() / 100:.2f})")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


x = torch.randn(M, Cc, device=dev); hb = torch.randn(M, hid, device=dev).to(BF); da2 = torch.randn(M, Cc, device=dev).to(BF)
lib = L.load()
for splits in (64, 128):
    for tag, dy, xx, N, K in (("fc1 dY bf16[.,512] x X f32[.,128]", ops.op_bf16(hb), ops.op_f32(x), hid, Cc),
                              ("fc2 dY bf16[.,128] x X gelu(bf16[.,512])", ops.op_bf16(da2), ops.op_bf16(hb, gelu=True), Cc, hid)):
        nb = lib.swv2_linear_wgrad_ws_bytes(M, N, K, splits)
        ws = torch.zeros(nb // 4, device=dev)
        dW = torch.zeros(N, K, device=dev)
        import ctypes as C
        f = lambda: L.check(lib.swv2_linear_wgrad_ws(C.byref(dy), C.byref(xx), dW.data_ptr(), None, None, None, K, splits, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream), "wgrad")
        t = timeit(f)
        report(f"{tag} splits={splits}", ws, nb // (128 * 128 * 4), t)
