export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/r04; cd /tmp
rm -rf /tmp/pa /tmp/pb /tmp/pc
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d /tmp/pa -o p --output-format csv -- python3 $R/tools/perf_probe.py attn > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pb -o p --output-format csv -- python3 $R/tools/perf_probe.py attn > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM -d /tmp/pc -o p --output-format csv -- python3 $R/tools/perf_probe.py attn > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pa","/tmp/pb","/tmp/pc"):
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "attn_fwd" not in k: continue
        import re
        m=re.search(r"(attn_\w+_kernel<[^>]*>)", k); k=m.group(1) if m else k[:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        print(k[:44], {c.replace('SQ_',''): round(x/1e6,1) for c,x in v.items()})
PY
