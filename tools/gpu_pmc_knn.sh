#!/bin/bash
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/knn_probe.py 1000000 > /dev/null 2>&1
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof/kn1 -o p -- python $R/tools/knn_probe.py 1000000 > $R/gpurun_out/prof/kn1.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof/kn1/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "pt2pl_tile" in k or "pt2pt_knn" in k:
        a = acc[(k[:40], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, t) in sorted(acc.items()):
    print(f"{k[0]:42s} {k[1]:22s} per dispatch {t / n:16.0f}")
PY
