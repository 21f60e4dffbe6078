#!/bin/bash
# tools/gpu_calib.sh: FETCH_SIZE / WRITE_SIZE of tools/probes/fetch_calib.bin (known bytes, the search kernels' access
# patterns), each counter in a pass of its own; summary -> gpurun_out/calib/summary.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calib; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
$R/tools/probes/fetch_calib.bin > $O/known_bytes.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $R/tools/probes/fetch_calib.bin > $O/f.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o p -- $R/tools/probes/fetch_calib.bin > $O/k.log 2>&1; echo "stats rc=$?"
cd $R
python - > $O/summary.txt <<PY
import csv, glob
print(open("$O/known_bytes.txt").read())
for f in glob.glob("$O/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": print("FETCH_SIZE raw (KB units) %-40s %12.0f  -> x1024 = %.1f MB, x2048 = %.1f MB" % (r["Kernel_Name"][:40], float(r["Counter_Value"]), float(r["Counter_Value"]) * 1024 / 1e6, float(r["Counter_Value"]) * 2048 / 1e6))
for f in glob.glob("$O/k/**/*kernel_stats.csv", recursive=True):
    for i, l in enumerate(open(f)):
        if i == 0 or "gather" in l or "stream" in l: print(l.strip()[:160])
PY
cat $O/summary.txt; rm -rf $O/f $O/k
