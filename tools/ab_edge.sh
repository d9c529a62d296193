R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_libs.sh "$1" --shape c3 --dtype bf16 --ranges 1 2>/dev/null | grep -v amdgpu
