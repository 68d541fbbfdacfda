#!/bin/bash
# Experiment build of the library with extra defines, next to the product one (never replaces it):
#   tools/build_variant.sh nt -DOFX_NT_STORE   ->  octfusion_amd/libofx_nt.so   (use with OFX_LIB=octfusion_amd/libofx_nt.so)
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
obj=octfusion_amd/csrc/_obj_$name
mkdir -p $obj
pids=()
for s in octfusion_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $s -o $obj/$(basename ${s%.hip}).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
echo "extern \"C\" const char* ofx_build_hash() { return \"variant-$name\"; } extern \"C\" int ofx_build_ablation() { return 0; }" > $obj/ofx_buildinfo.cpp
/opt/rocm/bin/hipcc -O2 -fPIC -c $obj/ofx_buildinfo.cpp -o $obj/ofx_buildinfo.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o octfusion_amd/libofx_$name.so $obj/*.o
echo octfusion_amd/libofx_$name.so
