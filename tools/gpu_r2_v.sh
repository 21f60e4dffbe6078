#!/bin/bash
# SQ issue / wait counters of the search kernels on mid-chain iterations (two PMC passes, kernel trace only)
mkdir -p gpurun_out/r2v; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2v
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p1 -o p -- python $R/tools/nn_one.py chain 6 > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/p2 -o p -- python $R/tools/nn_one.py chain 6 > $O/p2.log 2>&1; echo "p2 rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    fs = glob.glob(f"gpurun_out/r2v/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"][:40]
        if "nn_" not in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(d, k, {c: round(t / n) for c, (n, t) in v.items()})
PY
