#!/bin/bash
# builds a variant of the library beside the product one: scripts/build_variant.sh stamps -DGLIO_DEV_STAMPS
# -> glio_amd/lib/libglio_hip_<name>.so, to be selected with GLIO_HIP_LIB (the product library is not touched)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
obj=glio_amd/lib/obj_$name
mkdir -p $obj
pids=()
for s in glio_amd/csrc/*.hip; do
  o=$obj/$(basename ${s%.hip}).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value "$@" -c $s -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $obj/*.o -o glio_amd/lib/libglio_hip_$name.so
echo glio_amd/lib/libglio_hip_$name.so
