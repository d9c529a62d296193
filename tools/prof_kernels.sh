#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace of tools/bench_kernels.py for one shape -> compact per-kernel table.
#   tools/prof_kernels.sh <tag> <shape> [env assignments...]
tag=$1; shape=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -- python3 $R/tools/bench_kernels.py --shape $shape > $R/gpurun_out/$tag/run.log 2>&1
grep dtype $R/gpurun_out/$tag/run.log
python3 $R/tools/stats_csv.py $(find $R/gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1) 8
