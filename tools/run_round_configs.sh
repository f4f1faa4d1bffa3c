set -x
C="--cpu-baseline off --ref-seconds 0"
python bench.py $C --game "breakthrough(rows=6,columns=6)" --playouts 800 > gpurun_out/r3_bench_c3_f32x.json 2> gpurun_out/r3_c3a.err
python bench.py $C --game "breakthrough(rows=6,columns=6)" --playouts 800 --precision f16 > gpurun_out/r3_bench_c3_f16.json 2> gpurun_out/r3_c3b.err
python bench.py $C --game "breakthrough(rows=8,columns=8)" --playouts 1600 --blocks 20 --slots 2048 --steps 1 --warmup 1 --precision f16 > gpurun_out/r3_bench_c5_f16.json 2> gpurun_out/r3_c5b.err
python bench.py $C --game "breakthrough(rows=8,columns=8)" --playouts 1600 --blocks 20 --slots 2048 --steps 1 --warmup 1 > gpurun_out/r3_bench_c5_f32x.json 2> gpurun_out/r3_c5a.err
python bench.py $C --weights checkpoint > gpurun_out/r3_bench_c2_ckpt.json 2> gpurun_out/r3_c2c.err
python tools/generation_wallclock.py --precision f32x > gpurun_out/r3_generation_wallclock_f32x.json 2> gpurun_out/r3_gw.err
python tools/generation_wallclock.py --precision f16 > gpurun_out/r3_generation_wallclock_f16.json 2>> gpurun_out/r3_gw.err
python tools/generation_wallclock.py --precision f32x --games 500 --playouts 100 --blocks 5 > gpurun_out/r3_generation_wallclock_f32x_500.json 2>> gpurun_out/r3_gw.err
for b in 256 512 1024 2048 3072 4096; do python tools/net_microbench.py --precision f32x --boards $b 2>&1 | tail -1; done > gpurun_out/r3_tower_vs_boards.txt
for b in 256 512 1024 2048 4096; do python tools/net_microbench.py --precision f16 --boards $b 2>&1 | tail -1; done >> gpurun_out/r3_tower_vs_boards.txt
for f in gpurun_out/r3_bench_c*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', round(d['value'],1), d['roofline']['ms_per_launch'], d['roofline_tree']['ms_per_launch'], d['compactions'], round(d['engine_hbm_gb'],1))"; done
