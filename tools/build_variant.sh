#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]: a one-off library build into tools/ab/NAME.so (git-ignored) for same-box A/B runs
# through SC_LIB_PATH; e.g. tools/build_variant.sh fin_clocks -DSC_FIN_CLOCKS
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/ab/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
for s in kernels gkr api; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c sumcheck_amd/csrc/$s.hip -o tools/ab/obj_$name/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/ab/obj_$name/*.o -o tools/ab/$name.so
rm -rf tools/ab/obj_$name
echo tools/ab/$name.so
