# PMC passes of the image kernels (one counter group per pass, as the MI355X guide prescribes): bash tools/probes/pmc_disc.sh
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_disc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/tools/probes/disc_kernel_times.py --reps 1 > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tools/probes/disc_kernel_times.py --reps 1 > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt | head -120
