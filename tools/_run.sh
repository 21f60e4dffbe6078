export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O
export MP2P_HIP_TUNE=predict=0
pass() { ( cd /tmp && timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -o p -- python $R/tools/nn_one.py chain 6 > $O/$1.log 2>&1; echo "$1 rc=$?" ); }
pass t1 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
pass t2 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
pass t3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
pass t4 "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TCP_UTCL1_STALL_MULTI_MISS_sum"
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("t1","t2","t3","t4"):
    fs = glob.glob(f"gpurun_out/r3o/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d,"none"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"][:40]
        if "nn_" not in k or "reset" in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(d, k, {c: round(t / n) for c, (n, t) in v.items()})
PY
