#!/bin/bash
# A throwaway build of the library with extra flags for dmi_kernels.hip only (A/B measurements): build_variant.sh <suffix> <flags...>
# → draco-oxide_amd/libdraco_mi_<suffix>.so (git-ignored; travels with gpurun).  Run things against it with DMI_LIBRARY=<that file>.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/draco-oxide_amd/csrc
name=$1; shift
make -s -C "$src"
mkdir -p /tmp/dmi_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c -o /tmp/dmi_variants/k_$name.o "$src/dmi_kernels.hip"
objs=$(cd "$src" && ls *.o | grep -v '^dmi_kernels.o$' | sed "s|^|$src/|")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/draco-oxide_amd/libdraco_mi_$name.so" /tmp/dmi_variants/k_$name.o $objs -lpthread
echo "$root/draco-oxide_amd/libdraco_mi_$name.so"
