#!/bin/bash
# memory-side counters of the search kernels on mid-chain iterations (one --pmc pass per group, kernel trace only)
mkdir -p gpurun_out/r2aa; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2aa
cd /tmp
timeout 60 rocprofv3 -L > $O/avail.txt 2>&1 || timeout 60 rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -c . $O/avail.txt
pass() { timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -o p -- python $R/tools/nn_one.py chain 6 > $O/$1.log 2>&1; echo "$1 rc=$?"; }
pass m1 "MemUnitBusy MemUnitStalled LDSBankConflict L2CacheHit"
pass m2 "TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
pass m3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
pass m4 "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR"
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("m1", "m2", "m3", "m4"):
    fs = glob.glob(f"gpurun_out/r2aa/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"][:40]
        if "nn_" not in k or "reset" in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(d, k, {c: round(t / n, 2) for c, (n, t) in v.items()})
PY
