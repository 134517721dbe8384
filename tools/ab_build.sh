#!/bin/bash
# Dev tool: variant libraries under build/ab/ that differ from the product library in ONE object.
#   tools/ab_build.sh NAME SOURCE.hip "EXTRA FLAGS"   ->  build/ab/liblicos_NAME.so
set -e
name=$1; src=$2; extra=$3
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I$root/licos_amd/csrc -Wno-unused-function \
  -munsafe-fp-atomics -fno-slp-vectorize $extra -x hip -c $root/licos_amd/csrc/$src -o $root/build/ab/$name.o
objs=$(ls $root/build/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/ab/liblicos_$name.so $objs $root/build/ab/$name.o
