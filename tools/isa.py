#!/usr/bin/env python3
"""ISA of one kernel of a csrc file (build container): instruction mix, spills, loop structure -- what the compiler made of a loop.
usage: tools/isa.py attn.hip 'attn_bwd_kernel<11, 1, true, 162, 1, true>' [-o out.s] [-- extra hipcc flags]
Found with it in round 5: a scalar `switch` in a rolled loop lowered to ~24 register copies per step (attention backward with bias),
`__builtin_amdgcn_permlane32_swap`'s second result compiled as its first (loss epilogue)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra, out = [], None
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
if "-o" in args:
    i = args.index("-o")
    out = args[i + 1]
    del args[i:i + 2]
src, want = args[0], args[1]
path = src if os.path.exists(src) else os.path.join(ROOT, "swin_v2_weather_amd", "csrc", src)
tmp = "/tmp/swv2_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", tmp, path] + extra,
               check=True, stderr=subprocess.DEVNULL)
lines = open(tmp).read().splitlines()
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_Z") and "; @" in l]
names = subprocess.run(["/usr/bin/c++filt"] + [n for _, n in starts], stdout=subprocess.PIPE).stdout.decode().splitlines()
norm = lambda t: re.sub(r"\s+", "", t.replace("(anonymous namespace)::", ""))
hits = [(i, n) for (i, _), n in zip(starts, names) if norm(want) in norm(n)]
if not hits:
    print("no kernel matches; candidates:")
    print("\n".join(sorted({re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")) for n in names})))
    sys.exit(1)
i0, name = hits[0]
i1 = next(j for j in range(i0, len(lines)) if lines[j].lstrip().startswith(".amdhsa_kernel"))
body = lines[i0:i1]
if out:
    open(out, "w").write("\n".join(body) + "\n")
ins = [l.split()[0] for l in body if l.startswith("\t") and not l.lstrip().startswith((";", "."))]
mix = collections.Counter(ins)
print(re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")), f": {len(ins)} instructions")
print("  " + "  ".join(f"{k} {v}" for k, v in mix.most_common(24)))
print("  loops:", sum("Loop Header" in l for l in body), " scratch ops:", sum(k.startswith("scratch_") for k in ins),
      " barriers:", mix.get("s_barrier", 0), " mfma:", sum(k.startswith("v_mfma") for k in ins),
      " v_mov:", sum(k.startswith("v_mov") or k.startswith("v_accvgpr") for k in ins))
