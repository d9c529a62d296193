# Round 5: bf16 node GEMMs, 32-row kernel vs the LDS-staged 9-wave kernel (c3 shape = 256 ragged, c2 = 64 x 27)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_gemm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -m pytest $R/tests/test_hip_parity.py -q -x -k "bf16" 2>&1 | tail -5
for rep in 1 2; do for lds in 1 2; do for rg in 1 2; do
  echo -n "lds=$lds ranges=$rg  "; python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges $rg --bf16-lds $lds
done; done; done
for lds in 1 2; do echo -n "c2 lds=$lds  "; python3 $R/tools/bench_kernels.py --shape c2 --dtype bf16 --bf16-lds $lds; done
for lds in 1 2; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$lds -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --bf16-lds $lds > $O/prof$lds.log 2>&1
  f=$(find $O/prof$lds -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_bf16_lds${lds}_kernel_stats.csv
  rm -rf $O/prof$lds
  head -6 $O/c3_bf16_lds${lds}_kernel_stats.csv | cut -c1-150
done
