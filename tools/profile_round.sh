# Round evidence, run on the GPU box: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>/; copy what is quoted into profiles/)
set -x
R=$GRAFT_REPO_ROOT; T=${1:-round}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_c2.json 2> $O/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-x6-probe --no-config2 --no-config0 --no-config4 > $O/prof_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $O/pmc_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) > $O/pmc_$c.csv
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $O/pmc_mfma.log 2>&1
python3 $R/tools/pmc_summary.py $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1) > $O/pmc_mfma.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/pmc_insts -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $O/pmc_insts.log 2>&1
python3 $R/tools/pmc_summary.py $(find $O/pmc_insts -name "*counter_collection.csv" | head -1) > $O/pmc_insts.csv
# configs[2] shape (256 ragged molecules): the same two traffic passes + the instruction mix
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_c3_$c -- python3 $R/tools/bench_kernels.py --shape c3 --iters 1 > $O/pmc_c3_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $(find $O/pmc_c3_$c -name "*counter_collection.csv" | head -1) > $O/pmc_c3_$c.csv
done
python3 $R/bench.py --n-samples 256 --variance 12 --no-cpu-baseline --no-x6-probe > $O/bench_c3.json 2>/dev/null
python3 $R/bench.py --fragment --dtype bf16 --n-samples 256 --variance 12 --diffusion-steps 250 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_c5_share.json 2>/dev/null
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/tools/stats_csv.py $O/kernel_stats.csv 10
head -3 $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv | cut -c1-200
