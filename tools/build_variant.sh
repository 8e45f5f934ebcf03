#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]  -> tools/_build/libnerfsig_<name>.so (same sources, extra defines); select it with NERFSIG_LIB
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/_build/$name
mkdir -p $out
for f in nerf_signature_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -fno-gpu-rdc -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c $f -o $out/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/libnerfsig_$name.so $out/*.o
echo tools/_build/libnerfsig_$name.so
