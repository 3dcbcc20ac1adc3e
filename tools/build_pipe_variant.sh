#!/bin/bash
# tools/build_pipe_variant.sh <name> <extra hipcc flags...>: variants/libatomdev_<name>.so = the default library's objects
# with k_rk_atm_pipe.hip recompiled with the extra flags (A/B timing and stamp experiments of the pipelined sweep)
set -e
name=$1; shift
cd "$(dirname "$0")/../atom_amd/csrc"
make -s -j8
mkdir -p ../../variants/$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=fast --offload-arch=gfx950 -I../../include -I. "$@" -c k_rk_atm_pipe.hip -o ../../variants/$name/k_rk_atm_pipe.o
objs=$(ls *.o | grep -v '^k_rk_atm_pipe.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libatomdev_$name.so $objs ../../variants/$name/k_rk_atm_pipe.o
echo built variants/libatomdev_$name.so
