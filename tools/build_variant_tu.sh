#!/bin/bash
# Like tools/build_variant.sh, but rebuilds only the named translation units with the extra flags and links them with the
# in-tree objects of everything else (run `make -C raytracer_amd/csrc` first):
#   tools/build_variant_tu.sh NAME "kernels_path.hip kernels_tail.hip" "-DRTX_ABLATE_MISS"  ->  raytracer_amd/variants/librtx_NAME.so
set -e
name=$1; tus=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/raytracer_amd/csrc
out=$root/raytracer_amd/variants
tmp=$(mktemp -d)
mkdir -p "$out"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$src $*"
pids=()
for f in $tus; do
  /opt/rocm/bin/hipcc $FLAGS -c "$src/$f" -o "$tmp/${f%.*}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=""
for o in "$src"/*.o; do
  b=$(basename "$o")
  if [ -f "$tmp/$b" ]; then objs="$objs $tmp/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/librtx_$name.so" $objs -pthread
rm -rf "$tmp"
echo "built $out/librtx_$name.so"
