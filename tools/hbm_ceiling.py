#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU (GPU box): torch copy (1 read + 1 write per element), read-only reduction,
write-only fill, and an fp32 -> bf16 cast (the mixed traffic of the block kernels).  The numbers calibrate the roofline fractions:
peak stays the guide's 8 TB/s, these are the achievable rates beside it."""
import torch
dev = torch.device("cuda:0")
n = 1 << 28                                  # 1 GiB of fp32
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
yb = torch.empty(n, dtype=torch.bfloat16, device=dev)


def t(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e-3


for name, fn, nbytes in (("copy fp32 (read + write)", lambda: y.copy_(x), 8 * n), ("sum (read only)", lambda: x.sum(), 4 * n),
                         ("fill (write only)", lambda: y.fill_(1.0), 4 * n), ("cast fp32 -> bf16", lambda: yb.copy_(x), 6 * n),
                         ("add (2 reads + 1 write)", lambda: torch.add(x, y, out=y), 12 * n)):
    s = t(fn)
    print(f"{name:28s} {nbytes / s / 1e12:6.2f} TB/s  ({s * 1e6:.0f} us for {nbytes / 1e9:.2f} GB)")
