set -x
python -m pytest tests/test_fused_net.py -x -q -s -k "f32x or split" > gpurun_out/r2_t4.log 2>&1; echo rc=$? >> gpurun_out/r2_t4.log
grep -q "rc=0" gpurun_out/r2_t4.log && python -m pytest tests/test_precision_search_gpu.py -x -q -s > gpurun_out/r2_t4b.log 2>&1; echo rc=$? >> gpurun_out/r2_t4b.log
python bench.py --cpu-baseline off > gpurun_out/r2_bench_x3.json 2> gpurun_out/r2_bench_x3.err
bash tools/profile_round.sh gpurun_out/prof_r2 > gpurun_out/r2_prof.log 2>&1
