#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: builds variants/libatomdev_<name>.so for A/B timing
set -e
name=$1; shift
cd "$(dirname "$0")/../atom_amd/csrc"
mkdir -p ../../variants/$name
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I../../include -I."
for f in *.hip; do
  f=${f%.hip}
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o ../../variants/$name/$f.o &
done
for f in k_rk_atm_ws k_rk_hyd_ws; do
  grep -q ATOMDEV_RK_FAST $f.hip && /opt/rocm/bin/hipcc $FLAGS -ffp-contract=fast -DATOMDEV_RK_FAST "$@" -c $f.hip -o ../../variants/$name/${f}_fast.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libatomdev_$name.so ../../variants/$name/*.o
echo built variants/libatomdev_$name.so
