"""CPU oracle for the L2HMC leapfrog hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement (fp64 by default, fp32 on request) of the
algorithm the reference implements in TensorFlow-1.x graph ops.  It exists so
that the HIP kernels in ``l2hmc_amd/csrc`` can be checked against an
independent statement of the same mathematics on identical, explicitly
injected inputs (weights, masks, momenta, direction coins, MH uniforms).

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``l2hmc_amd/`` imports,
calls, links or executes anything from here; the product path raises if the
HIP library is missing instead of falling back to this code.

Pinning status -- **parity unpinned at tensor level**.  The reference ships no
tests, golden tensors or fixtures and its modules
all ``import tensorflow`` (1.x APIs), which is not installed in this image and
cannot be fetched, so the reference itself cannot be executed to produce
vectors.  What *is* pinned against the reference's own known answers
(tests/test_oracle_kat.py):

  (i)   <plaquette> = I1(beta)/I0(beta)       l2hmc/lattice/lattice.py:31-33
  (ii)  cold start => action 0, plaq 1, Q 0   notebooks/gauge_model_graph_mode.ipynb:255-265
  (iii) mask indices from np.random.seed(42)  dynamics/gauge_dynamics.py:651-661, globals.py:12
  (iv)  analytic force == autodiff of action  dynamics/gauge_dynamics.py:698-709 vs
                                               lattice/gauge_lattice.py:427-459
  (v)   backward_lf o forward_lf == identity, sum-log-det == log|det J|,
        hmc=True degenerates to plain leapfrog gauge_dynamics.py:102-108,537-590

Modules: ``lattice`` / ``nets`` / ``gauge_dynamics`` / ``dynamics`` / ``loss`` (NumPy, forward values),
``torch_ref`` (the same graphs in float64 torch ops so that torch.autograd stands in for tf.gradients -- the
checker for the training kernels; pinned to the NumPy modules, to finite differences and to the train_*.npz
fixtures by tests/test_oracle_train.py), ``stats`` (lag-sum autocorrelations), ``cpu_baseline`` (torch-CPU
fp32 op-for-op port, the timed baseline of bench.py).

Every function cites the reference file:line it restates (paths relative to
``/root/reference``).
"""
