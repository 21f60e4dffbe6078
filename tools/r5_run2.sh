#!/bin/bash
out=gpurun_out/r5b; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python tools/ab_probe.py $out/ab.json \
  "sel_lane:nn_direct=0" \
  "sel_lane_cap12k:nn_direct=0,tile_cand_cap=12288" \
  "sel_lane_cap24k:nn_direct=0,tile_cand_cap=24576" \
  "sel_lane_cap100k:nn_direct=0,tile_cand_cap=100000" \
  "sel_lane_cap24k_coop0:nn_direct=0,tile_cand_cap=24576,coop_max=0" \
  "sel_lane_cap24k_hard1000:nn_direct=0,tile_cand_cap=24576,hard_cand=1000" \
  "sel_lane_cap24k_hard2500:nn_direct=0,tile_cand_cap=24576,hard_cand=2500" \
  "sel_lane_cap24k_easy6k:nn_direct=0,tile_cand_cap=24576,tile_cand_cap_easy=6144" \
  > $out/ab.txt 2> $out/ab.err
echo "ab rc=$?" | tee -a $out/rc.txt
cat $out/ab.txt
