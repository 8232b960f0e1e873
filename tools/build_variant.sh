#!/bin/bash
# build a library variant for A/B timing inside ONE gpurun call: tools/build_variant.sh <name> "<extra hipcc flags>"
# -> fluidsolvergpu_amd/libsfgpu_<name>.so (select it with SF_LIB=libsfgpu_<name>.so; see tools/exp_cmp.sh, bench_kernels.sh)
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/fluidsolvergpu_amd/csrc; obj=/tmp/sfvar_$name; mkdir -p $obj
flags="-O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -Wno-unused-function $extra"
for u in sf_api sf_solver_f32 sf_solver_f64; do /opt/rocm/bin/hipcc $flags -c -o $obj/$u.o $src/$u.hip & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/fluidsolvergpu_amd/libsfgpu_$name.so $obj/sf_api.o $obj/sf_solver_f32.o $obj/sf_solver_f64.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built libsfgpu_$name.so
