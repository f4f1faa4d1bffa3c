set -x
python -m pytest tests/test_overlap_gpu.py tests/test_full_size_other_configs_gpu.py "tests/test_fused_net.py" -x -q > gpurun_out/r2_t2.log 2>&1; echo rc=$? >> gpurun_out/r2_t2.log
B="python bench.py --cpu-baseline off --ref-seconds 0"
$B > gpurun_out/r2_c2_base.json 2> gpurun_out/r2_c2_base.err
$B --overlap 2 > gpurun_out/r2_c2_ov2.json 2> gpurun_out/r2_c2_ov2.err
$B --overlap 2 --slots 8192 > gpurun_out/r2_c2_ov2_8192.json 2> gpurun_out/r2_c2_ov2_8192.err
$B --slots 8192 > gpurun_out/r2_c2_8192.json 2> gpurun_out/r2_c2_8192.err
$B --game "breakthrough(rows=6,columns=6)" --playouts 800 > gpurun_out/r2_c3.json 2> gpurun_out/r2_c3.err
$B --game "breakthrough(rows=6,columns=6)" --playouts 800 --overlap 2 > gpurun_out/r2_c3_ov2.json 2> gpurun_out/r2_c3_ov2.err
$B --game "breakthrough(rows=8,columns=8)" --playouts 1600 --blocks 20 --slots 2048 --steps 1 > gpurun_out/r2_c5.json 2> gpurun_out/r2_c5.err
$B --game "breakthrough(rows=8,columns=8)" --playouts 1600 --blocks 20 --slots 2048 --steps 1 --overlap 2 > gpurun_out/r2_c5_ov2.json 2> gpurun_out/r2_c5_ov2.err
echo finished
