# kernel-level view of tools/probes/interp_rate.py: bash tools/probes/prof_interp.sh [planes [size]]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_interp_${2:-1024}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/tools/probes/interp_rate.py ${1:-128} ${2:-1024} > $OUT/run.log 2>&1
cat $OUT/stats/*/*kernel_stats.csv | cut -c1-200 | head -30
