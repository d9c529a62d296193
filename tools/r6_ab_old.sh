R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_libs.sh "base oldedge" --shape c3 --dtype bf16 --ranges 1 --edge-pair 1 > $O/ab_old.txt 2>&1
bash $R/tools/ab_libs.sh "base oldedge" --shape c3 --dtype bf16 --ranges 2 --edge-pair 1 >> $O/ab_old.txt 2>&1
bash $R/tools/ab_libs.sh "base oldedge" --shape c2 --dtype f32x6 >> $O/ab_old.txt 2>&1
cat $O/ab_old.txt | grep -v "amdgpu.ids: No"
