#!/bin/bash
# A/B of SQ counters for the GCL edge kernel with / without workgroup-level sums (separate --pmc passes): tools/pmc_ab.sh <tag>
R=$GRAFT_REPO_ROOT; tag=${1:-ab}
cd /tmp && export TMPDIR=/tmp
for wg in 1 0; do
  export MCG_WG_SUMS=$wg
  for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
    t=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$tag/wg${wg}_$t -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/$tag/wg${wg}_$t -name "*counter_collection.csv" | head -1) | grep -E "k_edge_lds<1, false" | sed "s/^/wg_sums=$wg /" | cut -c1-30,100-200
  done
done
