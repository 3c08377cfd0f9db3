#!/bin/bash
# Developer tool: build a variant of the library with extra -D flags on gemm.hip into abtmp/lib<name>.so (git-ignored; it
# travels to the GPU box).  usage: tools/build_variant.sh <name> [hipcc flags...]   then  SYN3R_LIB_OVERRIDE=abtmp/lib<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p abtmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -fno-gpu-rdc "$@" -c syn3r_amd/csrc/${SRC:-gemm}.hip -o abtmp/${SRC:-gemm}_$name.o
objs=$(ls syn3r_amd/build/*.o | grep -v "/${SRC:-gemm}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o abtmp/lib$name.so abtmp/${SRC:-gemm}_$name.o $objs
echo abtmp/lib$name.so
