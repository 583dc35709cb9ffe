#!/usr/bin/env python3
"""Register / LDS / spill summary per kernel of one csrc file (build container; hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py attn2.hip [name-filter] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
src = args[0]
flt = args[1] if len(args) > 1 else ""
path = src if os.path.exists(src) else os.path.join(ROOT, "swin_v2_weather_amd", "csrc", src)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", path, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPR Spill: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
if not rows:
    print(out[-3000:])
    sys.exit(1)
names = subprocess.run(["/usr/bin/c++filt"] + [r["name"] for r in rows], stdout=subprocess.PIPE).stdout.decode().splitlines()
print(f"{'vgpr':>5s} {'agpr':>5s} {'spill':>5s} {'scr':>5s} {'lds':>7s} {'occ':>3s}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    if flt and flt not in n:
        continue
    print(f"{r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('spill', 0):5d} {r.get('scratch', 0):5d} {r.get('lds', 0):7d} {r.get('occ', 0):3d}  {n}")
