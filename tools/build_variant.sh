#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: builds gpurun_out/../variants/libatomdev_<name>.so for A/B timing
set -e
name=$1; shift
cd "$(dirname "$0")/../atom_amd/csrc"
mkdir -p ../../variants/$name
for f in atomdev_api k_pointwise k_rk_atm k_rk_atm_ws k_rk_hyd k_bc_pressure k_rk2d k_accuracy k_columns k_transfer k_ice_scheme k_results_hyd k_vtk k_ic_hyd k_read_atm k_slice_hyd k_slice_atm; do
  [ -f $f.hip ] || continue
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. "$@" -c $f.hip -o ../../variants/$name/$f.o &
done
for f in k_rk_hyd_ws; do [ -f $f.hip ] && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I. "$@" -c $f.hip -o ../../variants/$name/$f.o & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libatomdev_$name.so ../../variants/$name/*.o
echo built variants/libatomdev_$name.so
