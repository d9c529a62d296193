#!/bin/bash
# Measurement builds of libmlconfgen_hip.so with extra -D flags on the EGNN translation units, e.g.
#   tools/build_variants.sh noremap -DMCG_NO_XCD_REMAP  precise -DMCG_PRECISE=2
# -> tools/native/variants/libmlconfgen_hip_<tag>.so ; run with MCG_LIB_PATH=<that file>.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/ml_conformer_generator_amd/csrc"
OUT="$ROOT/tools/native/variants"
mkdir -p "$OUT"
make -C "$CS" -j4 >/dev/null          # the shared objects of the untouched sources
VAR="mcg_edge_exact mcg_edge_bf16 mcg_egnn_api"
REST="mcg_egnn_model.o mcg_egnn_plan.o mcg_plan_host.o mcg_sampler.o mcg_gcn.o mcg_misc.o mcg_shape.o mcg_post.o mcg_devmem.o"
while [ $# -ge 2 ]; do
  tag="$1"; flags="$2"; shift 2
  objs=""
  for f in $VAR; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$CS/$f.hip" -o "$OUT/${f}_$tag.o" &
    objs="$objs $OUT/${f}_$tag.o"
  done
  wait
  ( cd "$CS" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmlconfgen_hip_$tag.so" $objs $REST )
  rm -f $objs
  echo "built $tag ($flags)"
done
