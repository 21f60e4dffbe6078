#!/bin/bash
out=gpurun_out/r5h; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python tools/ab_probe.py $out/ab.json "nosplit:split_cand=0" "empty16k:split_cand=100000000,split_tiles=16" "empty64k:split_cand=100000000,split_tiles=64" "empty128k:split_cand=100000000,split_tiles=120" "nosplit_noxcd:split_cand=0,xcd_map=0" > $out/ab.txt 2> $out/ab.err
echo "ab rc=$?" | tee -a $out/rc.txt
cat $out/ab.txt
