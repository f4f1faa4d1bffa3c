#!/bin/bash
# Re-takes the small per-round evidence files on the GPU box (copy what is wanted from gpurun_out/ into profiles/).
set -x
export TMPDIR=/tmp
C="--cpu-baseline off --ref-seconds 0 --companion-steps 0"
for b in 256 512 1024 2048 3072 4096; do python tools/net_microbench.py --precision f32x --boards $b 2>&1 | tail -1; done > gpurun_out/r3_tower_vs_boards.txt
for b in 256 512 1024 2048 4096; do python tools/net_microbench.py --precision f16 --boards $b 2>&1 | tail -1; done >> gpurun_out/r3_tower_vs_boards.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3k -- python3 bench.py $C --game "breakthrough(rows=6,columns=6)" --playouts 800 --precision f16 --steps 2 --warmup 1 > gpurun_out/c3k.json 2> gpurun_out/c3k.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5k -- python3 bench.py $C --game "breakthrough(rows=8,columns=8)" --playouts 1600 --blocks 20 --slots 2048 --precision f16 --steps 1 --warmup 1 > gpurun_out/c5k.json 2> gpurun_out/c5k.err
find gpurun_out/c3k gpurun_out/c5k -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo refreshed
