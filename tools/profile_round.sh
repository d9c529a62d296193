# Round evidence, run on the GPU box: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>/; copy what is quoted into profiles/)
# Every step runs under its own timeout; rocprofv3 gets the program itself after `--` (never a shell or env wrapper).
set -x
R=$GRAFT_REPO_ROOT; T=${1:-round}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-x6-probe --no-config2 --no-config0 --no-config4 > $O/prof_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_$c.csv
done
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $O/pmc_mfma.log 2>&1
f=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_mfma.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/pmc_insts -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $O/pmc_insts.log 2>&1
f=$(find $O/pmc_insts -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_insts.csv
# configs[2] shape (256 ragged molecules): the same two traffic passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_c3_$c -- python3 $R/tools/bench_kernels.py --shape c3 --iters 1 > $O/pmc_c3_$c.log 2>&1
  f=$(find $O/pmc_c3_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_c3_$c.csv
done
# configs[4] arithmetic (bf16 operands) at the configs[2] shape: instruction mix + matrix-pipe busy cycles of the 64-row edge kernel
# (the issue-bound yardstick of bench.py's config4 object), one molecule range so that kernels do not overlap
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS --output-format csv -d $O/pmc_c3_bf16_insts -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --iters 1 > $O/pmc_c3_bf16_insts.log 2>&1
f=$(find $O/pmc_c3_bf16_insts -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_c3_bf16_insts.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_c3_bf16_mfma -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --iters 1 > $O/pmc_c3_bf16_mfma.log 2>&1
f=$(find $O/pmc_c3_bf16_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_c3_bf16_mfma.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_bf16 -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 > $O/prof_c3_bf16.log 2>&1
f=$(find $O/prof_c3_bf16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_bf16_kernel_stats.csv
timeout 400 python3 $R/bench.py --n-samples 256 --variance 12 --no-cpu-baseline --no-x6-probe > $O/bench_c3.json 2>/dev/null
timeout 400 python3 $R/bench.py --fragment --dtype bf16 --n-samples 256 --variance 12 --diffusion-steps 250 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5_share.json 2>/dev/null
# multi-GPU code paths on the one GPU there is: the real RCCL collectives on a 1-rank group, and the 2-rank control flow over gloo
MCG_FORCE_COLLECTIVE=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --no-x6-probe --no-config2 --no-config0 --no-config4 > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
MCG_DIST_BACKEND=gloo MCG_BENCH_TIMEOUT=500 timeout 600 python3 $R/bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo_dryrun_one_gpu.json 2> $O/bench_2rank_gloo.err
timeout 120 python3 $R/tools/rccl_one_rank_check.py > $O/rccl_one_rank_check.json 2> $O/rccl_one_rank_check.err
f=$(find $O/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
python3 $R/tools/stats_csv.py $O/kernel_stats.csv 10
head -3 $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv | cut -c1-200
