#!/bin/bash
# On the GPU box, from the repo root: the 0.25-degree measurements (single GPU, and two slabs sharing the card as a rehearsal).
set -u
R=$PWD
mkdir -p gpurun_out/refined
timeout -k 10 300 python -u bench.py --grid 0.25 --steps 40 --warmup 6 --no-cpu > gpurun_out/refined/bench_quarter.json 2> gpurun_out/refined/bench_quarter.err || { echo "bench 0.25 failed"; tail -5 gpurun_out/refined/bench_quarter.err; exit 1; }
cat gpurun_out/refined/bench_quarter.json
timeout -k 10 300 python -u bench.py --grid 0.25 --mode slabs --steps 40 --warmup 6 > gpurun_out/refined/bench_quarter_slab1.json 2> gpurun_out/refined/slab1.err || { echo "slab x1 failed"; tail -5 gpurun_out/refined/slab1.err; exit 1; }
cat gpurun_out/refined/bench_quarter_slab1.json
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --mode slabs --grid 0.25 --backend gloo --steps 20 --warmup 4 > gpurun_out/refined/bench_quarter_slab2_one_card.json 2> gpurun_out/refined/slab2.err || { echo "slab x2 failed"; tail -8 gpurun_out/refined/slab2.err; exit 1; }
cat gpurun_out/refined/bench_quarter_slab2_one_card.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/refined/prof -o quarter -- python3 $R/bench.py --grid 0.25 --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/refined/rocprof.log 2>&1 || { echo "rocprof failed"; tail -5 $R/gpurun_out/refined/rocprof.log; exit 1; }
cd $R
f=$(ls gpurun_out/refined/prof/*kernel_stats.csv gpurun_out/refined/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
head -12 $f | cut -d, -f1-8
