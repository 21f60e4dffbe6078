#!/bin/bash
# tools/gpu_pmc.sh <tag> "<probe command>" <kernel-name substring> [stats] [sq] [mem] [timeline]
#   SQ issue / wait counters, memory-side counters and the kernel-stats table of ONE probe command, e.g.
#     tools/gpu_pmc.sh k3 "tools/nn_one.py chain 6" nn_ stats sq mem        (pt2pt search kernels, mid-chain poses)
#     tools/gpu_pmc.sh k5 "tools/pl_one.py 120000"  pt2pl_ stats sq mem     (point-to-plane search + fit)
#   every counter group is a pass of its own, never together with a runtime trace; raw output under gpurun_out/<tag>/
tag=$1; cmd=$2; filt=$3; shift 3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
pass() { ( cd /tmp && timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -o p -- python $R/$cmd > $O/$1.log 2>&1; echo "$1 rc=$?" ); }
for step in "$@"; do
  case $step in
    timeline) timeout 300 python tools/timeline_probe.py > $O/timeline.json 2> $O/timeline.err; echo "timeline rc=$?"; cat $O/timeline.json;;
    sq1) pass p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY";;
    sq) pass p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"
        pass p2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
        pass p3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES_EQ_64";;
    mem) pass m1 "FETCH_SIZE"; pass m2 "WRITE_SIZE"; pass m3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum";;
    stats) ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o p -- python $R/$cmd > $O/ks.log 2>&1; echo "stats rc=$?" );;
  esac
done
cd $R
python - > $O/summary.txt <<PY
import csv, glob, collections
for d in ("p1","p2","p3","m1","m2","m3"):
    fs = glob.glob(f"gpurun_out/$tag/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"][:${KLEN:-44}]
        if "$filt" not in k or "reset" in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(d, k, {c: round(t / n) for c, (n, t) in v.items()})
for f in glob.glob(f"gpurun_out/$tag/ks/**/*kernel_stats.csv", recursive=True):
    for i,l in enumerate(open(f)):
        if i == 0 or "mp2p::" in l: print(l.strip()[:200])
PY
cat $O/summary.txt
# the raw per-dispatch CSVs of a python probe run to tens of MB (torch's own kernels): only the summary and the
# kernel-stats table travel back (gpurun merges at most 64 MiB)
mkdir -p $O/keep; cp $O/ks/*kernel_stats.csv $O/keep/ 2>/dev/null; rm -rf $O/p1 $O/p2 $O/p3 $O/m1 $O/m2 $O/m3 $O/ks
