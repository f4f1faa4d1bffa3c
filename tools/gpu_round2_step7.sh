set -x
B="python bench.py --cpu-baseline off"
$B --weights checkpoint > gpurun_out/r2_c2_ckpt.json 2> gpurun_out/r2_c2_ckpt.err
$B --ref-seconds 0 --precision f32x --game "breakthrough(rows=6,columns=6)" --playouts 800 --steps 1 > gpurun_out/r2_c3_f32x.json 2> gpurun_out/r2_c3_f32x.err
$B --ref-seconds 0 --precision f32x --game "breakthrough(rows=8,columns=8)" --playouts 1600 --blocks 20 --slots 2048 --steps 1 --warmup 0 > gpurun_out/r2_c5_f32x.json 2> gpurun_out/r2_c5_f32x.err
$B --ref-seconds 0 --weights checkpoint --game "breakthrough(rows=6,columns=6)" --playouts 800 > gpurun_out/r2_c3_ckpt.json 2> gpurun_out/r2_c3_ckpt.err
bash tools/profile_round.sh gpurun_out/prof_r2b > gpurun_out/r2_profb.log 2>&1
echo finished
