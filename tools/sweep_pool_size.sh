# node-pool capacity against throughput (C3: breakthrough 6x6, 800 sims, 4096 slots, f16)
C="--cpu-baseline off --ref-seconds 0 --precision f16 --game breakthrough(rows=6,columns=6) --playouts 800"
for n in ${@:-120000 240000 430000}; do
  python bench.py $C --nodes-per-slot $n > gpurun_out/r3_c3_pool_$n.json 2> gpurun_out/r3_c3_pool.err
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r3_c3_pool_$n.json') if l.startswith('{')][-1]); print($n, round(d['value'],1), d['compactions'], round(d['engine_hbm_gb'],1), d['roofline_tree']['ms_per_launch'])"
done
