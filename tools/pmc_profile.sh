#!/bin/bash
# Profile bench.py on the GPU box: (1) rocprofv3 --kernel-trace --stats, (2) PMC counters in
# separate passes (counters only; never combined with sys/hip/hsa tracing). Writes under
# gpurun_out/pmc_<tag>/ and prints a summary. Usage: bash tools/pmc_profile.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
ARGS="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras $ARGS > $OUT/bench_stats.log 2>&1 || echo "stats pass failed"
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --preheat-steps 0 --no-cpu-baseline --no-extras $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run wr WRITE_SIZE
run rd FETCH_SIZE
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt
cat $OUT/stats/*/*kernel_stats.csv | head -8
# (copy $OUT/traffic.json to profiles/traffic.json for the headline frame, profiles/traffic_<workload>.json otherwise:
#  bench.py reports it as roofline.traffic while the library's sha256 matches)
grep '^{' $OUT/bench_stats.log | tail -1 | cut -c1-200
