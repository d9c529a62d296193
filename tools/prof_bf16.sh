set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_bf16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS --output-format csv -d $O/pmc_insts -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --iters 1 > $O/pmc_insts.log 2>&1
f=$(find $O/pmc_insts -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_c3_bf16_insts.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --iters 1 > $O/pmc_mfma.log 2>&1
f=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/pmc_c3_bf16_mfma.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_bf16_kernel_stats.csv
timeout 400 python3 $R/bench.py --fragment --dtype bf16 --n-samples 256 --variance 12 --diffusion-steps 250 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5_share.json 2>/dev/null
timeout 400 python3 $R/bench.py > $O/bench_c2.json 2>/dev/null
