set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 -m pytest $R/tests -m gpu -x -q -k "eight_wave" > $O/pytest_pair.log 2>&1; echo "rc=$?" >> $O/pytest_pair.log
tail -5 $O/pytest_pair.log
echo "# pair=2 (eight-wave workgroups), c3 bf16, one range" > $O/ab.txt
bash $R/tools/ab_libs.sh "base nodeph d2 d5 nob nobar noa" --shape c3 --dtype bf16 --ranges 1 --edge-pair 2 >> $O/ab.txt 2>&1
echo "# pair=1 (four-wave workgroups), c3 bf16, one range" >> $O/ab.txt
bash $R/tools/ab_libs.sh "base nob nobar noa" --shape c3 --dtype bf16 --ranges 1 --edge-pair 1 >> $O/ab.txt 2>&1
grep -v amdgpu.ids $O/ab.txt
