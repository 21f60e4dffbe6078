#!/bin/bash
# the N > 1 flow of bench.py on ONE GPU (two ranks sharing it, gloo for the collectives): exercises the
# sharded step, the max-over-ranks timing, the strong-scaling block and the JSON line; numbers are meaningless
mkdir -p gpurun_out/r2ah; export TMPDIR=/tmp
MP2P_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2ah/bench_n2.json 2> gpurun_out/r2ah/bench_n2.err; echo "n2 rc=$?"
tail -c 1500 gpurun_out/r2ah/bench_n2.json; tail -3 gpurun_out/r2ah/bench_n2.err | cut -c1-300
