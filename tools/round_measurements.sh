#!/bin/bash
# The measurements a round's numbers come from, on one box: the default command under rocprof + PMC (tools/profile_round.sh), the
# default command plain, configs 3 and 5 at the default precision.  tools/round_measurements.sh PREFIX  -> gpurun_out/PREFIX_*
P=${1:-r4}
C="--cpu-baseline off --ref-seconds 0"
set -x
timeout -k 10 600 tools/profile_round.sh gpurun_out/${P}prof || exit 1
timeout -k 10 300 python3 bench.py > gpurun_out/${P}_bench_default.json 2> gpurun_out/${P}_bench_default.err || exit 1
tail -c 600 gpurun_out/${P}_bench_default.json
timeout -k 10 200 python3 bench.py $C --game "breakthrough(rows=6,columns=6)" --playouts 800 > gpurun_out/${P}_bench_c3_f32x.json 2> gpurun_out/${P}_c3.err || exit 1
timeout -k 10 400 python3 bench.py $C --game "breakthrough(rows=8,columns=8)" --playouts 1600 --blocks 20 --slots 2048 --steps 1 --warmup 1 > gpurun_out/${P}_bench_c5_f32x.json 2> gpurun_out/${P}_c5.err || exit 1
for f in gpurun_out/${P}_bench_*.json; do python3 -c "
import json
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', round(d['value'],1), d['roofline']['ms_per_launch'], d['roofline_tree']['ms_per_launch'], d['roofline'].get('frac'), d['roofline'].get('traffic'))"; done
