#!/bin/bash
# A/B build of the library (never shipped / never loaded by default): tools/build_variant.sh NAME "-DFLAG=..." ->
# tools/_diag/lib_NAME.so, loaded through L2HMC_LIB_PATH.  Objects that do not depend on the flags are reused.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../l2hmc_amd/csrc"
OUT=../../tools/_diag/var_$NAME
mkdir -p $OUT
for f in capi u1_lattice stq_dense leapfrog small_mlp fused_traj fused_traj4 fused_traj32 conv3d_front mcmc_step loss train small_train fused_train; do
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on "$@" -c $f.hip -o $OUT/$f.o || touch $OUT/FAILED ) &
done
wait
if [ -e $OUT/FAILED ]; then rm -f $OUT/FAILED; echo "variant build FAILED"; exit 1; fi
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o ../../tools/_diag/lib_$NAME.so
echo built tools/_diag/lib_$NAME.so
