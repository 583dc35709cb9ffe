#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of `bench.py --force-ddp` -> which compute kernels ran while each RCCL kernel was in flight.
usage: ddp_overlap.py <trace dir> <out.md>"""
import csv
import glob
import os
import sys


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]


def main(d, out):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
    ev.sort()
    is_cc = lambda n: any(t in n.lower() for t in ("nccl", "rccl", "allreduce", "all_reduce"))
    cc = [e for e in ev if is_cc(e[2])]
    comp = [e for e in ev if not is_cc(e[2])]
    lines = ["# Gradient all-reduce vs backward kernels (rocprofv3 --kernel-trace of `bench.py --force-ddp`, one-rank RCCL group)", "",
             f"{len(ev)} kernel records, {len(cc)} of them RCCL; queues seen: {sorted({e[3] for e in ev})}", ""]
    if not cc:
        lines.append("No RCCL kernel in the trace: with one rank RCCL completes the all-reduce without launching a device kernel "
                     "(the in-place one-rank collective is a no-op), so overlap cannot be shown on a 1-GPU box.")
    else:
        lines += ["| RCCL kernel | start (us into trace) | duration us | queue | compute kernels in flight during it (queue) |", "|---|---|---|---|---|"]
        t0 = ev[0][0]
        n_over = 0
        for s, e, n, q, st in cc[-40:]:
            during = sorted({f"{short(k[2])} (q{k[3]})" for k in comp if k[0] < e and k[1] > s})
            n_over += bool(during)
            lines.append(f"| `{short(n)}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {q} | {', '.join(during) if during else '-'} |")
        lines += ["", f"{n_over} of the last {min(len(cc), 40)} RCCL kernels overlap at least one compute kernel."]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
