#!/bin/bash
# tools/build_variant1.sh <name> <file.hip> <extra hipcc flags...>: variants/libatomdev_<name>.so with one source file
# rebuilt with the extra flags and every other object taken from the in-tree build (A/B timing of one kernel)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../atom_amd/csrc"
mkdir -p ../../variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I."
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o ../../variants/${name}_${src%.hip}.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libatomdev_$name.so $objs ../../variants/${name}_${src%.hip}.o
echo built variants/libatomdev_$name.so
