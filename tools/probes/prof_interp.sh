set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_interp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/tools/probes/interp_rate.py 128 > $OUT/run.log 2>&1
cat $OUT/stats/*/*kernel_stats.csv | cut -c1-200 | head -30
