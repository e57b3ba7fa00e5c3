#!/bin/bash
# Collect PMC counters for bench.py (separate passes; counters only, no tracing domains
# besides --kernel-trace). Usage on the GPU box: bash tools/pmc_profile.sh <tag>
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq3 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
run wr WRITE_SIZE
run rd FETCH_SIZE
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt
