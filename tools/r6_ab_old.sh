R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_libs.sh "base nodeph" --shape c3 --dtype bf16 --ranges 1 --edge-pair 2 > $O/ab.txt 2>&1
bash $R/tools/ab_libs.sh "base oldedge" --shape c3 --dtype bf16 --ranges 1 --edge-pair 1 >> $O/ab.txt 2>&1
cat $O/ab.txt | grep -v "amdgpu.ids: No"
