#!/bin/bash
# tools/build_fast_variant.sh <name> atm|hyd <extra flags...>: variants/libatomdev_<name>.so with the tolerance-form
# object of one sweep rebuilt with the extra flags
set -e
name=$1; which=$2; shift 2
cd "$(dirname "$0")/../atom_amd/csrc"
mkdir -p ../../variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=fast --offload-arch=gfx950 -I../../include -I. -DATOMDEV_RK_FAST"
/opt/rocm/bin/hipcc $FLAGS "$@" -c k_rk_${which}_ws.hip -o ../../variants/${name}_fast.o
objs=$(ls *.o | grep -v "^k_rk_${which}_ws_fast.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libatomdev_$name.so $objs ../../variants/${name}_fast.o
echo built variants/libatomdev_$name.so
