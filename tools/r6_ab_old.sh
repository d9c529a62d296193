R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 -m pytest $R/tests -m gpu -x -q -k "bf16 or f32x6 or x6 or config5 or randomized or split_operand" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash $R/tools/ab_libs.sh "base oldedge" --shape c3 --dtype bf16 --ranges 1 > $O/ab.txt 2>&1
bash $R/tools/ab_libs.sh "base oldedge" --shape c3 --dtype bf16 >> $O/ab.txt 2>&1
bash $R/tools/ab_libs.sh "base oldedge" --shape c2 --dtype f32x6 >> $O/ab.txt 2>&1
bash $R/tools/ab_libs.sh "base oldedge" --mols 64 --dtype bf16 >> $O/ab.txt 2>&1
grep -v "amdgpu.ids: No" $O/ab.txt
