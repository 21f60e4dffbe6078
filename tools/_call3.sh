#!/bin/bash
O=gpurun_out/${1:-r6_p1}; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/c5_single_probe.py 2>$O/err.txt | tee $O/single.txt; tail -2 $O/err.txt
