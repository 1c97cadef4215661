#!/bin/bash
# Build libcoloc_hip.so with extra compiler flags into tools/bin/<name>.so (experiments: COLOC_HIP_LIB=tools/bin/<name>.so python3 ...)
# usage: tools/build_variant.sh <name> [flags ...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/bin
cd coloc_amd/csrc
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -ldl "$@" -o ../../tools/bin/$NAME.so \
  capi.hip k2nn.hip clatch.hip lerp.hip pnp.hip detect.hip acransac.hip multicam.hip
echo built tools/bin/$NAME.so
