#!/bin/bash
# full GPU suite + smoke, then bench lines and kernel traces (nopos, relpos)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
pick='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"],1), "samples/s", d["step_ms"]["sequence"][:3], d.get("secondary"))'
python bench.py --no-cpu-baseline 2>$O/b0.err | tee $O/bench_full.json | python -c "$pick" main+secondary
bash tools/trace_bench.sh r05_relpos --rel-pos 1 --no-secondary > $O/trace_relpos.txt 2>&1
bash tools/trace_bench.sh r05_nopos --no-secondary > $O/trace_nopos.txt 2>&1
cp gpurun_out/r05_relpos_kernel_stats.md gpurun_out/r05_nopos_kernel_stats.md $O/
