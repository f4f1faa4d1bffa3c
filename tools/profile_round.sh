#!/bin/bash
# rocprofv3 evidence for a bench command (run on the GPU box; copy the summaries into profiles/).
#   tools/profile_round.sh OUT [bench.py flags ...]      default flags: none = the driver's default command (f32x)
# Counters go in their own passes (8 SQ slots; FETCH_SIZE and WRITE_SIZE cannot share a pass); no trace domains beside --pmc.
# The program comes directly after `--` (no env / bash -c hop: the profiler's library has initialised the GPU by then).
set -x
export TMPDIR=/tmp
( while true; do date +%T >> ${1:-gpurun_out/prof}.heartbeat; sleep 45; done ) &   # the box kills a command that is silent for 7 minutes
HB=$!
trap "kill $HB" EXIT
OUT=${1:-gpurun_out/prof}
shift
FLAGS="$@"
COMMON="--cpu-baseline off --ref-seconds 0"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $COMMON $FLAGS > $OUT.trace_bench.json 2> $OUT.trace.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq1 -- python3 bench.py $COMMON --steps 1 --warmup 0 --tick-limit 1024 --no-graph $FLAGS > $OUT.sq1.json 2> $OUT.sq1.err
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/sq2 -- python3 bench.py $COMMON --steps 1 --warmup 0 --tick-limit 1024 --no-graph $FLAGS > $OUT.sq2.json 2> $OUT.sq2.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- python3 bench.py $COMMON --steps 1 --warmup 0 --tick-limit 1024 --no-graph $FLAGS > $OUT.fetch.json 2> $OUT.fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -- python3 bench.py $COMMON --steps 1 --warmup 0 --tick-limit 1024 --no-graph $FLAGS > $OUT.write.json 2> $OUT.write.err
python3 - $OUT.trace_bench.json "$FLAGS" > $OUT.summary.txt <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("# command: python3 bench.py --cpu-baseline off --ref-seconds 0 %s   (under rocprofv3; PMC passes: --steps 1 --warmup 0 --tick-limit 1024 --no-graph: counter collection stalls on graph replays)" % sys.argv[2])
print("# workload_key: " + d["config"]["workload_key"])
print("# bench line under --kernel-trace: %.1f games/s, forward %.4f ms (HIP events), tick kernel %.4f ms"
      % (d["value"], d["roofline"]["ms_per_launch"], d["roofline_tree"]["ms_per_launch"]))
PY
python3 tools/pmc_summary.py $OUT/trace $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write >> $OUT.summary.txt 2>&1
rm -rf $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write   # raw per-dispatch CSVs are large; the summary is what is kept
find $OUT/trace -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo profiled
