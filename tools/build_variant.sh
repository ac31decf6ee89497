#!/bin/bash
# Builds an experimental variant of librtx_amd.so with extra compile flags, for A/B runs on the GPU box:
#   tools/build_variant.sh NAME "-DSHADE_WAVES_PER_EU=3"   ->  raytracer_amd/variants/librtx_NAME.so
# PATH_FLAGS="..." adds flags to kernels_path.hip only (the shading translation unit); PATH_DIV=-fno-hip-fp32-correctly-rounded-divide-sqrt
# rebuilds it with the 2.5-ulp division / square root that round 1 shipped.
# Select it at run time with RTX_AMD_LIB=raytracer_amd/variants/librtx_NAME.so (see raytracer_amd/_ffi.py).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/raytracer_amd/csrc
out=$root/raytracer_amd/variants
tmp=$(mktemp -d)
mkdir -p "$out"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$src $*"
pids=()
for f in kernels_trace.hip kernels_path.hip kernels_nee.hip kernels_tail.hip kernels_legacy.hip kernels_photon.hip kernels_post.hip kernels_sort.hip rtx_noalloc.hip rtx_api.cpp rtx_render.cpp rtx_photon.cpp rtx_group.cpp bvh_build.cpp; do
  extra=""
  [ "$f" = kernels_path.hip ] && extra="$PATH_DIV $PATH_FLAGS"
  /opt/rocm/bin/hipcc $FLAGS $extra -c "$src/$f" -o "$tmp/${f%.*}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/librtx_$name.so" "$tmp"/*.o -pthread
rm -rf "$tmp"
echo "built $out/librtx_$name.so"
