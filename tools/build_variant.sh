#!/bin/bash
# A/B builds of the library with different macros of one source (default gemm.hip): tools/build_variant.sh NAME "-DVAE_X3_PIPE=0 ..." [SRC]
# -> vae_amd/lib/variants/libvae_amd_NAME.so (select with VAE_AMD_LIB=...; the other objects come from build/)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; SRC=${3:-gemm}
mkdir -p build vae_amd/lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c vae_amd/csrc/$SRC.hip -o build/${SRC}_$NAME.o
OBJS=""; for o in runtime gemm rowops likelihood epilogue middle step optim data; do [ "$o" != "$SRC" ] && OBJS="$OBJS build/$o.o"; done   # (never another variant's object)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build/${SRC}_$NAME.o -o vae_amd/lib/variants/libvae_amd_$NAME.so
echo built vae_amd/lib/variants/libvae_amd_$NAME.so
