# Round 5: stall attribution of the configs[4] bf16 edge kernel (and, for contrast, the exact-fp32 one).
# PMC passes only (--kernel-trace + --pmc; the program itself right behind `--`).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_stall; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name dtype shape counters...
  n=$1; dt=$2; sh=$3; shift 3
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/bench_kernels.py --shape $sh --dtype $dt --ranges 1 --iters 1 > $O/$n.log 2>&1
  f=$(find $O/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/$n.csv
  rm -rf $O/$n
}
pass c3_bf16_stall_a bf16 c3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass c3_bf16_stall_b bf16 c3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
pass c3_bf16_stall_c bf16 c3 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_REQ GRBM_GUI_ACTIVE
pass c3_bf16_stall_d bf16 c3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
pass c2_f32_stall_a f32 c2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass c2_f32_stall_b f32 c2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_bf16_kernel_stats.csv
rm -rf $O/prof
tail -3 $O/*.log
