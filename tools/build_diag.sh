#!/bin/bash
# Diagnostic build of the library with in-kernel cycle stamps (never shipped / never loaded by l2hmc_amd).
set -e
cd "$(dirname "$0")/../l2hmc_amd/csrc"
OUT=../../tools/_diag
mkdir -p $OUT
for f in capi u1_lattice stq_dense leapfrog small_mlp fused_traj; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DL2HMC_STAMPS -c $f.hip -o $OUT/$f.o
done
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $OUT/libl2hmc_hip_diag.so
echo built $OUT/libl2hmc_hip_diag.so
