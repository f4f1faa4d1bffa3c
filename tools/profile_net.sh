#!/bin/bash
# rocprofv3 counters for the PV-net forward alone (tools/net_microbench.py).  tools/profile_net.sh OUT [net_microbench flags ...]
# (environment, e.g. AZ_NET_TOWER=x3b, is exported by the caller: the program itself comes right after `--`)
export TMPDIR=/tmp
OUT=${1:-gpurun_out/netprof}
shift
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/net_microbench.py "$@" > $OUT.trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq1 -- python3 tools/net_microbench.py "$@" > $OUT.sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/sq2 -- python3 tools/net_microbench.py "$@" > $OUT.sq2.log 2>&1
python3 tools/pmc_summary.py $OUT/trace $OUT/sq1 $OUT/sq2 > $OUT.summary.txt 2>&1
rm -rf $OUT/sq1 $OUT/sq2
find $OUT/trace -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo profiled
