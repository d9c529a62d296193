#!/bin/bash
# Measurement builds of libmlconfgen_hip.so with extra -D flags, e.g.
#   tools/build_variants.sh noremap -DMCG_NO_XCD_REMAP  precise -DMCG_PRECISE=2
# -> tools/native/variants/libmlconfgen_hip_<tag>.so ; run with MCG_LIB_PATH=<that file>.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/ml_conformer_generator_amd/csrc"
OUT="$ROOT/tools/native/variants"
mkdir -p "$OUT"
make -C "$CS" -j4 >/dev/null          # the shared objects of the untouched sources
pids=()
while [ $# -ge 2 ]; do
  tag="$1"; flags="$2"; shift 2
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c "$CS/mcg_egnn.hip" -o "$OUT/mcg_egnn_$tag.o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmlconfgen_hip_$tag.so" "$OUT/mcg_egnn_$tag.o" \
        "$CS/mcg_sampler.o" "$CS/mcg_gcn.o" "$CS/mcg_misc.o" "$CS/mcg_shape.o" "$CS/mcg_post.o"
    rm -f "$OUT/mcg_egnn_$tag.o"
    echo "built $tag ($flags)"
  ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
