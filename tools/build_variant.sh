#!/bin/bash
# A/B builds of the library with different gemm.hip macros: tools/build_variant.sh NAME "-DVAE_X3_PIPE=0 ..."
# -> vae_amd/lib/variants/libvae_amd_NAME.so (select with VAE_AMD_LIB=...; the other objects come from build/)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
mkdir -p build vae_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c vae_amd/csrc/gemm.hip -o build/gemm_$NAME.o
OBJS=$(ls build/*.o | grep -v "build/gemm")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build/gemm_$NAME.o -o vae_amd/lib/variants/libvae_amd_$NAME.so
echo built vae_amd/lib/variants/libvae_amd_$NAME.so
