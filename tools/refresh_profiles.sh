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
echo refreshed kernel stats
# wall clock of the reference-facing calls, and the kernels of the reference Trainer's own generation size (500 games x 100 playouts, 5 blocks)
python tools/generation_wallclock.py --precision f32x > gpurun_out/r3_generation_wallclock_f32x.json 2> gpurun_out/r3_gw.err
python tools/generation_wallclock.py --precision f16 > gpurun_out/r3_generation_wallclock_f16.json 2>> gpurun_out/r3_gw.err
python tools/generation_wallclock.py --precision f32x --games 500 --playouts 100 --blocks 5 > gpurun_out/r3_generation_wallclock_f32x_500.json 2>> gpurun_out/r3_gw.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g500 -- python3 tools/generation_wallclock.py --precision f32x --games 500 --playouts 100 --blocks 5 > gpurun_out/g500.json 2> gpurun_out/g500.err
find gpurun_out/g500 -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo refreshed wallclock
