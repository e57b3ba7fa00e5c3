# What k_reproject<double> moves on BASELINE config 5 (512 planes of 1024^2 onto the 1 deg map), counted three ways:
# FETCH_SIZE / WRITE_SIZE, the L2's fabric read requests split by size (TCC_EA0_RDREQ, _32B), L2 hits / misses.
# bash tools/probes/pmc_reproject.sh  -> gpurun_out/pmc_reproject/summary.txt (the footprint of the map at 16 .. 256-byte
# granularity is arithmetic: tools/probes/reproject_footprint.py, CPU)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_reproject
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --workload cube --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1 || echo "stats pass failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/rd -- $B > $OUT/rd.log 2>&1 || echo "rd failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/wr -- $B > $OUT/wr.log 2>&1 || echo "wr failed"
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/rq -- $B > $OUT/rq.log 2>&1 || echo "rq failed"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- $B > $OUT/l2.log 2>&1 || echo "l2 failed"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || echo "sq failed"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt; grep -A14 "k_reproject" $OUT/summary.txt | head -40
grep k_reproject $OUT/stats/*/*kernel_stats.csv | head -3
