#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]: a one-off library build into tools/ab/NAME.so (git-ignored) for same-box A/B runs
# through SC_LIB_PATH (tools/ab.sh); e.g. tools/build_variant.sh fin_clocks -DSC_FIN_CLOCKS
# REUSE="kernels_big gkr" takes those translation units' objects from the production build (sumcheck_amd/build/, which must be current)
# instead of compiling them again -- for flags that only touch the others (kernels_big.hip alone is two minutes of hipcc).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/ab/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for s in kernels_big kernels kernels_tail kernels_wide kernels_wide16 gkr abi protocol comm; do
  case " $REUSE " in
    *" $s "*) cp sumcheck_amd/build/$s.hip.o tools/ab/obj_$name/$s.o;;
    *) /opt/rocm/bin/hipcc $FLAGS "$@" -c sumcheck_amd/csrc/$s.hip -o tools/ab/obj_$name/$s.o &;;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ab/obj_$name/*.o -o tools/ab/$name.so
rm -rf tools/ab/obj_$name
echo tools/ab/$name.so
