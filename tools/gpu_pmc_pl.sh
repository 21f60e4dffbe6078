#!/bin/bash
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/pl_one.py 1000000 > /dev/null 2>&1
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof/pl1 -o p -- python $R/tools/pl_one.py 1000000 > $R/gpurun_out/prof/pl1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof/pl2 -o p -- python $R/tools/pl_one.py 1000000 > $R/gpurun_out/prof/pl2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("pl1", "pl2"):
    f = glob.glob(f"gpurun_out/prof/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print("no output for", d); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if "pt2pl_tile" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, t) in sorted(acc.items()):
        print(f"{k:24s} per dispatch {t / n:16.0f}")
PY
