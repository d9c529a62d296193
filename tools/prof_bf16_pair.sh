# Round 6: the eight-wave / 128-row bf16 edge workgroup (MCG_OPT_EDGE_BF16_PAIR = 2) against the four-wave one (= 1) at the 256-ragged
# shape, one molecule range: kernel trace + the stall counters of round 5.  PMC passes only (--kernel-trace + --pmc).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_pair_stall; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name pair counters...
  n=$1; pr=$2; shift 2
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --iters 1 --edge-pair $pr > $O/$n.log 2>&1
  f=$(find $O/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/$n.csv
  rm -rf $O/$n
}
for pr in 1 2; do
  pass pair${pr}_stall_a $pr SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  pass pair${pr}_stall_b $pr SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
  pass pair${pr}_stall_c $pr TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_REQ GRBM_GUI_ACTIVE
  pass pair${pr}_stall_d $pr SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$pr -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --edge-pair $pr > $O/prof$pr.log 2>&1
  f=$(find $O/prof$pr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/pair${pr}_kernel_stats.csv
  rm -rf $O/prof$pr
done
tail -2 $O/*.log
