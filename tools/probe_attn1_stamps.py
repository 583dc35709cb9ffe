#!/usr/bin/env python3
"""In-kernel phase timing of the two-phase attention backward (attn_bwd_kernel): private build with -DSWV2_ATTN1_STAMPS;
wave 0 of the workgroups of head 0 sums s_memtime deltas per phase -- GPU box, diagnostics only."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L
# a prebuilt stamp library (tools/build_variant_all.sh stamps "-DSWV2_ATTN1_STAMPS -DSWV2_ATTN2_STAMPS") or a build on the box
so = os.path.join(ROOT, "swin_v2_weather_amd", "libswv2_stamps.so")
if not os.path.exists(so):
    so = "/tmp/libswv2_a1stamps.so"
    srcs = [os.path.join(L.CSRC, s) for s in L.SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSWV2_ATTN1_STAMPS",
                           "-DSWV2_ATTN2_STAMPS", "-o", so] + srcs, stderr=subprocess.DEVNULL)
L.LIB_PATH = so
# (the DMA-staged variant, tools/experiments/attn_bwd_dma.hip, is not in the library: DMA = 1 needs a build that links it)
DMA = os.environ.get("SWV2_ATTN_BWD_DMA", "0") != "0" and not (len(sys.argv) > 1 and sys.argv[1] == "bias")
from swin_v2_weather_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
B = int(os.environ.get("PROBE_B", "2"))
HD = int(os.environ.get("PROBE_HD", "16"))            # head width (24: BASELINE configs[4])
plan = ops.window_plan(B, 180, 360, 9, 18, 4, 9, 8, HD, 0)
Bw, h, Lp, DP, Lw = plan.Bw, 8, plan.Lp, plan.DP, plan.L
qkvh = (torch.randn(Bw, h, 3, Lp, DP, device=dev) * 0.25).to(BF); qkvh[:, :, :, Lw:] = 0
oh = torch.empty(Bw, h, Lp, DP, dtype=BF, device=dev); lse = torch.zeros(Bw, h, Lp, device=dev)
ls = torch.full((h,), 2.3, device=dev)
rel_pos = len(sys.argv) > 1 and sys.argv[1] == "bias"          # CPB bias variant (one key tile per wave, d bias in registers)
bias = torch.randn(h, Lw, Lw, device=dev) if rel_pos else None
dbias = torch.zeros(h, Lw, Lw, device=dev) if rel_pos else None
pk = ops.attn_pack_bias(bias) if rel_pos else None
a = ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, HD, plan.nwh, plan.nww, plan.mask_thr, max_chunks=32 if rel_pos else 64, bias_pack=pk)
ops.attn_fwd(a)
doh = torch.randn(Bw, h, Lp, DP, device=dev).to(BF); rnorm = torch.rand(Bw, h, 2, Lp, device=dev) + 0.5
dq, dls = torch.empty_like(qkvh), torch.zeros(h, device=dev)
a = ops.attn_args(qkvh, ls, bias, oh, lse, Bw, h, Lw, HD, plan.nwh, plan.nww, plan.mask_thr, doh=doh, rnorm=rnorm, dqkvh=dq, dlogit=dls,
                  dbias=dbias, max_chunks=32 if rel_pos else 256 // h, bias_pack=pk)
for _ in range(3):
    ops.attn_bwd(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_bwd(a)
e1.record(); torch.cuda.synchronize()
buf = torch.zeros(512 * 8, dtype=torch.int64)
assert getattr(ctypes.CDLL(so), "swv2_debug_attn2_stamps" if DMA else "swv2_debug_attn1_stamps")(ctypes.c_void_p(buf.data_ptr())) == 0
allw = buf.view(512, 8).double()
nwg = 512 // 11
perwave = allw[:nwg * 11].view(nwg, 11, 8)
st = perwave[:, 0, :]
if DMA:
    print("DMA-staged kernel (attn_bwd2.hip): 'commit' = wait for the next window's DMA, 'barrier 3' = its statistics")
names = ["issue next window's prefetch", "phase 1 (S, dP, softmax bwd, dV, dK)", "dK / dV normalisation + stores", "barrier 1",
         "phase 2 (dQ) + stores", "barrier 2", "commit (prefetch wait, LDS, delta)", "barrier 3"]
tot = st.sum(1)
print(f"attn_bwd (two-phase): {e0.elapsed_time(e1) * 100:.1f} us; wave 0 of {len(st)} workgroups, total ticks mean {tot.mean():.0f}")
for i, n in enumerate(names):
    print(f"  {n:40s} {100 * st[:, i].mean() / tot.mean():5.1f} %  (min {st[:, i].min():.0f} max {st[:, i].max():.0f})")
print("per wave (mean ticks over the sampled workgroups): phase 1 | dK/dV + stores | barrier 1 | phase 2 | barrier 2")
for w in range(11):
    m = perwave[:, w, :].mean(0)
    print(f"  wave {w:2d}: {m[1]:8.0f} {m[2]:8.0f} {m[3]:8.0f} {m[4]:8.0f} {m[5]:8.0f}   issue {m[0]:7.0f} commit {m[6]:7.0f}")
