# round 6: eight-wave / 128-row bf16 edge workgroups - parity test, then A/B per launch and per call at several batch sizes
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 -m pytest $R/tests -m gpu -x -q -k "eight_wave or bf16_mode_vs_emulation or gathers_partial or randomized_batch or config5" > $O/pytest_pair.log 2>&1; echo "rc=$?" >> $O/pytest_pair.log
tail -5 $O/pytest_pair.log
for pair in 1 2; do
  for r in 1 2; do
    timeout 200 python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges $r --edge-pair $pair >> $O/ab_pair.txt 2>&1
  done
  for mols in 64 96 128 192; do
    timeout 200 python3 $R/tools/bench_kernels.py --mols $mols --dtype bf16 --ranges 1 --edge-pair $pair >> $O/ab_pair.txt 2>&1
  done
done
grep -v amdgpu.ids $O/ab_pair.txt
