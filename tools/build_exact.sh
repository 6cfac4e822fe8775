#!/bin/bash
# Diagnostic build (never shipped / never loaded by l2hmc_amd): the fused kernels with libm expf / tanhf in place
# of the v_exp_f32 / v_rcp_f32 forms, for tools/error_ratio.py --lib tools/_diag/libl2hmc_hip_exact.so
set -e
cd "$(dirname "$0")/../l2hmc_amd/csrc"
OUT=../../tools/_diag/exact
mkdir -p $OUT
for f in capi u1_lattice stq_dense leapfrog small_mlp fused_traj conv3d_front mcmc_step loss train small_train fused_train; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -DL2HMC_EXACT_MATH -c $f.hip -o $OUT/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o ../../tools/_diag/libl2hmc_hip_exact.so
echo built tools/_diag/libl2hmc_hip_exact.so
