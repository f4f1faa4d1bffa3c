set -x
python -m pytest tests/test_guards_gpu.py tests/test_engine_parity.py tests/test_replay.py tests/test_multirank_gpu.py -x -q > gpurun_out/r2_t3.log 2>&1; echo rc=$? >> gpurun_out/r2_t3.log
python tools/make_engine_export_fixture.py gpurun_out/engine_export_connect_four.npz > gpurun_out/r2_fixture.log 2>&1
bash tools/profile_round.sh gpurun_out/prof_r2 > gpurun_out/r2_prof.log 2>&1
