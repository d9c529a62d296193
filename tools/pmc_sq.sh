# SQ issue/wait counters of the edge kernels (separate --pmc passes, kernel-trace only): tools/pmc_sq.sh c3 f32
set -x
R=$GRAFT_REPO_ROOT; SHAPE=${1:-c3}; DT=${2:-f32}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/sq_${SHAPE}_${DT}_$tag -- python3 $R/tools/bench_kernels.py --shape $SHAPE --dtype $DT --iters 2 > /dev/null 2>&1
done
for d in $R/gpurun_out/sq_${SHAPE}_${DT}_*; do python3 $R/tools/pmc_summary.py $d/*/*_counter_collection.csv | grep -E "k_edge" ; done
