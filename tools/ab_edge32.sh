R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_libs.sh "$1" --shape c2 --dtype f32 2>/dev/null | grep -v amdgpu
