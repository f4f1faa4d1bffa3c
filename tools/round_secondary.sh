#!/bin/bash
# The round's secondary artefacts on one box: generation wall clock, forward time against batch size, host issue overhead, counters of
# the forward alone for the three board shapes.  tools/round_secondary.sh PREFIX -> gpurun_out/PREFIX_*
P=${1:-r4}
set -x
timeout -k 10 200 python3 tools/generation_wallclock.py --precision f32x > gpurun_out/${P}_generation_wallclock_f32x.json 2> gpurun_out/${P}_gw.err || exit 1
timeout -k 10 100 python3 tools/generation_wallclock.py --precision f32x --games 500 --playouts 100 --blocks 5 > gpurun_out/${P}_generation_wallclock_f32x_500.json 2>> gpurun_out/${P}_gw.err || exit 1
for b in 256 512 1024 1280 2048 3072 4096; do timeout -k 10 100 python3 tools/net_microbench.py --precision f32x --boards $b 2>&1 | tail -1; done > gpurun_out/${P}_tower_vs_boards.txt
timeout -k 10 300 python3 tools/host_issue_overhead.py > gpurun_out/${P}_host_issue_overhead.json 2> gpurun_out/${P}_hio.err || exit 1
timeout -k 10 200 tools/profile_net.sh gpurun_out/${P}_np_c4 --game connect_four --blocks 10 --boards 4096 --precision f32x --iters 20 || exit 1
timeout -k 10 200 tools/profile_net.sh gpurun_out/${P}_np_bt6 --game "breakthrough(rows=6,columns=6)" --blocks 10 --boards 4096 --precision f32x --iters 20 || exit 1
timeout -k 10 200 tools/profile_net.sh gpurun_out/${P}_np_bt8 --game "breakthrough(rows=8,columns=8)" --blocks 20 --boards 2048 --precision f32x --iters 20 || exit 1
echo done
