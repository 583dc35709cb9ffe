#!/usr/bin/env python3
"""Condense rocprofv3 output into the small summaries committed under profiles/.

  python profiles/summarize.py stats <dir with *_kernel_stats.csv> <steps profiled> <out.md>
  python profiles/summarize.py pmc   <dir with FETCH pass> <dir with WRITE pass> <out.json>
  python profiles/summarize.py counters <dir with PMC passes> <out.json>
  python profiles/summarize.py mfma  <sq counters json> <out.json>
  python profiles/summarize.py check <kernel_stats.md> <pmc_hbm.json> <pmc_mfma.json>     # same kernels in all three, or exit 1
"""
import csv
import glob
import json
import os
import sys


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]


def stats(d, steps, out):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({steps} profiled steps, total GPU time {tot / 1e6 / steps:.2f} ms/step)", "",
             "| kernel | launches/step | avg us | ms/step | % |", "|---|---|---|---|---|"]
    for r in rows[:40]:
        lines.append(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                     f"{float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['Percentage']):.1f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))


def pmc(dfetch, dwrite, out):
    res = {}
    for d, key in ((dfetch, "FETCH_SIZE"), (dwrite, "WRITE_SIZE")):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        acc = {}
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != key:
                continue
            n = short(r["Kernel_Name"])
            a = acc.setdefault(n, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for n, (cnt, val) in acc.items():
            res.setdefault(n, {})[key] = {"launches": cnt, "mean_kb": val / cnt}
    for n, v in res.items():
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE reads exactly 1/2 of a wide coalesced stream on gfx950; WRITE_SIZE exact; unit KB
        fe, wr = v.get("FETCH_SIZE", {}).get("mean_kb", 0.0), v.get("WRITE_SIZE", {}).get("mean_kb", 0.0)
        v["hbm_bytes_per_launch"] = (2.0 * fe + wr) * 1024.0
    res["_source_hash"] = source_hash()            # bench.py reports `traffic` only for the kernel sources these counters came from
    res["_local_batch"] = int(os.environ.get("SWV2_PROFILE_BATCH", 2))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for n in sorted((k for k in res if not k.startswith("_")), key=lambda k: -res[k]["hbm_bytes_per_launch"])[:12]:
        print(f"{res[n]['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  {n}")


def source_hash():
    """sha256 over the kernel sources + the C header (the same function as swin_v2_weather_amd._lib.source_hash)"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "swin_v2_weather_amd", "csrc", "*"))) + [os.path.join(root, "include", "swv2.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# the kernels bench.py reports roofline objects for (name fragments)
ROOFLINE_FRAGMENTS = ("attn_bwd_", "attn_fwd3_kernel", "mlp_bwd_kernel", "mlp_fwd_kernel", "gemm_tn_slab_c128_kernel")


def check(stats_md, hbm_json, mfma_json):
    """the three summaries of a round must talk about the SAME kernel instantiations (round 2 shipped a derived MFMA file for
    the kernel variant before the last commit): for every roofline kernel exactly one name per file, identical across files"""
    import re
    names = {"stats": set(re.findall(r"^\| `([^`]+)`", open(stats_md).read(), flags=re.M)),
             "hbm": {k for k in json.load(open(hbm_json)) if not k.startswith("_")},
             "mfma": set(json.load(open(mfma_json))["kernels"])}
    bad = []
    for frag in ROOFLINE_FRAGMENTS:
        per = {src: sorted(n for n in ns if frag in n) for src, ns in names.items()}
        flat = {tuple(v) for v in per.values()}
        if len(flat) != 1 or any(len(v) != 1 for v in per.values()):
            bad.append((frag, per))
    for frag, per in bad:
        print(f"MISMATCH {frag}: {per}")
    if bad:
        sys.exit(1)
    print("ok: the roofline kernels carry the same names in", stats_md, hbm_json, mfma_json)


def counters(root, out):
    """mean per-launch value of every counter of every pass directory under `root` -> {kernel: {counter: mean}}"""
    res = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for (n, c), (cnt, val) in acc.items():
            res.setdefault(n, {})[c] = val / cnt
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for n, v in sorted(res.items()):
        if "SQ_BUSY_CYCLES" in v or "SQ_WAVE_CYCLES" in v:
            print(n)
            for c in sorted(v):
                print(f"    {c:28s} {v[c]:16.0f}")


def mfma(sq_json, out):
    """counter-based MFMA pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (sum over the 1024 SIMDs of the cycles the
    matrix pipe was busy) / (SQ_BUSY_CYCLES x 32): SQ_BUSY_CYCLES is summed over the 32 shader engines (4 per XCD), so
    SQ_BUSY_CYCLES / 32 is the kernel's duration in shader clocks and x 1024 SIMDs gives the SIMD-cycles available."""
    d = json.load(open(sq_json))
    res = {}
    for n, v in d.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or v.get("SQ_INSTS_MFMA", 0) <= 0:
            continue
        simd_cycles = v["SQ_BUSY_CYCLES"] * 32.0
        res[n] = {"SQ_INSTS_MFMA": v["SQ_INSTS_MFMA"], "SQ_VALU_MFMA_BUSY_CYCLES": v["SQ_VALU_MFMA_BUSY_CYCLES"],
                  "SQ_BUSY_CYCLES": v["SQ_BUSY_CYCLES"], "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"),
                  "SQ_ACTIVE_INST_VALU": v.get("SQ_ACTIVE_INST_VALU"), "SQ_WAVES": v.get("SQ_WAVES"),
                  "cycles_per_mfma": v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_INSTS_MFMA"],
                  "kernel_shader_clocks": v["SQ_BUSY_CYCLES"] / 32.0,
                  "mfma_pipe_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                  "valu_issue_frac": (v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0) / simd_cycles}
    json.dump({"_formula": mfma.__doc__, "_source_hash": source_hash(), "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for n in sorted(res, key=lambda k: -res[k]["SQ_VALU_MFMA_BUSY_CYCLES"])[:14]:
        print(f"{res[n]['mfma_pipe_busy_frac']:6.3f} mfma  {res[n]['valu_issue_frac']:6.3f} valu  {n}")


if __name__ == "__main__":
    if sys.argv[1] == "check":
        check(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
