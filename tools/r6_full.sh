set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1500 python3 -m pytest $R/tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
for r in 1 2 3; do timeout 200 python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges $r 2>&1 | grep -v amdgpu.ids >> $O/ranges.txt; done
for mols in 128 160 192; do for r in 1 2; do timeout 200 python3 $R/tools/bench_kernels.py --mols $mols --dtype bf16 --ranges $r 2>&1 | grep -v amdgpu.ids >> $O/ranges.txt; done; done
cat $O/ranges.txt
