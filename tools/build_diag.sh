#!/bin/bash
# Diagnostic build of the library with in-kernel cycle stamps (never shipped / never loaded by l2hmc_amd).
set -e
cd "$(dirname "$0")/../l2hmc_amd/csrc"
for NW in ${DIAG_WAVES_LIST:-4}; do
  OUT=../../tools/_diag/w$NW
  mkdir -p $OUT
  for f in capi u1_lattice stq_dense leapfrog small_mlp fused_traj fused_traj4 fused_traj32 conv3d_front mcmc_step loss train small_train fused_train; do
    ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DL2HMC_STAMPS -DL2HMC_FUSED_WAVES=$NW $EXTRA -c $f.hip -o $OUT/$f.o || touch $OUT/FAILED ) &
  done
  wait
  if [ -e $OUT/FAILED ]; then rm -f $OUT/FAILED; echo "diagnostic build FAILED (see the compiler output above)"; exit 1; fi
  hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o ../../tools/_diag/libl2hmc_hip_diag_w$NW.so
done
cp ../../tools/_diag/libl2hmc_hip_diag_w4.so ../../tools/_diag/libl2hmc_hip_diag.so
echo built tools/_diag/libl2hmc_hip_diag_w*.so
