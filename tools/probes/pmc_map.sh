# PMC passes of the map kernels: bash tools/probes/pmc_map.sh
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_map
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/tools/probes/map_plane_rate.py > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tools/probes/map_plane_rate.py > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
