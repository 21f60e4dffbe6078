#!/bin/bash
out=gpurun_out/r5v; mkdir -p $out
export TMPDIR=/tmp
for v in "default:" "oldbudget:tile_cand_cap=6144,coop_max=4" "r4:tile_select=0" "cap6144:tile_cand_cap=6144" "coop4:coop_max=4" "direct:nn_direct=1" "direct_old:nn_direct=1,tile_cand_cap=6144,coop_max=4"; do
  name=${v%%:*}; knobs=${v#*:}
  MP2P_HIP_TUNE=$knobs timeout 300 python bench.py --config c2 --steps 40 --warmup 5 2>$out/c2_$name.err | grep '^{"metric"' > $out/c2_$name.json
  python -c "
import json; d=json.loads(open('$out/c2_$name.json').read()); print('c2 $name', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms search', round(d['kernel_ms']['search_last_matcher'],3))"
done
