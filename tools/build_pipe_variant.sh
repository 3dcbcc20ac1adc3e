#!/bin/bash
# tools/build_pipe_variant.sh <name> [atm|hyd] <extra hipcc flags...>: variants/libatomdev_<name>.so = the default library's
# objects with k_rk_atm_pipe.hip (or k_rk_hyd_pipe.hip) recompiled with the extra flags (A/B timing and stamp experiments of
# the pipelined sweeps)
set -e
name=$1; shift
src=k_rk_atm_pipe
if [ "$1" = hyd ]; then src=k_rk_hyd_pipe; shift; elif [ "$1" = atm ]; then shift; fi
cd "$(dirname "$0")/../atom_amd/csrc"
make -s -j8
mkdir -p ../../variants/$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=fast --offload-arch=gfx950 -I../../include -I. "$@" -c $src.hip -o ../../variants/$name/$src.o
objs=$(ls *.o | grep -v "^$src.o\$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libatomdev_$name.so $objs ../../variants/$name/$src.o
echo built variants/libatomdev_$name.so
