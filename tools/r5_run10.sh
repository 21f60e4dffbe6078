#!/bin/bash
out=gpurun_out/r5u; mkdir -p $out
export TMPDIR=/tmp
nproc > $out/cpu.txt; cat /sys/fs/cgroup/cpu.max >> $out/cpu.txt 2>/dev/null; lscpu | grep -i "model name\|socket\|core(s)\|thread(s)\|numa" >> $out/cpu.txt
python - >> $out/cpu.txt 2>&1 <<PY
import time, numpy as np, oracle as orc
from mp2p_icp_amd import synthetic
d = synthetic.make_scan_union_pair(1_000_000, 10_000_000, 1, map_scan_points=1_000_000)
g,l=d['glob'],d['local']
t0=time.time(); tree=orc.KDTree(g[:,0],g[:,1],g[:,2]); print('build',round(time.time()-t0,2))
ls=l[::5]
for th in (1,2,4,8,16,32,64,128,256,256):
    t0=time.time(); p,_=orc.match_pt2pt(g[:,0],g[:,1],g[:,2],ls[:,0],ls[:,1],ls[:,2],d['T_gt'],2.0,0.0,tree=tree,threads=th); t1=time.time()-t0
    print(th,'threads: 200k queries at T_gt', round(t1*1e3,1),'ms', 'pairs',len(p))
PY
cat $out/cpu.txt
for c in c2 c3 c5; do timeout 400 python bench.py --config $c --steps 40 --warmup 5 2>$out/$c.err | grep '^{"metric"' > $out/$c.json; python -c "
import json; d=json.loads(open('$out/$c.json').read()); print('$c', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d.get('kernel_ms'))"; done
