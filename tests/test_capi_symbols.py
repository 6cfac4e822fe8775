"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, exports every symbol include/l2hmc_hip.h declares, and rejects bad
arguments on the host (no kernel is launched in this file)."""
import ctypes as C
import os

import pytest
import torch

from l2hmc_amd import _lib, build as lbuild


@pytest.fixture(scope="module")
def L():
    lbuild.build()            # hipcc cross-compiles without a GPU; building is not a fallback
    return _lib.lib()


def test_library_exports_every_declared_symbol(L):
    declared = _lib.declared_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/l2hmc_hip.h but not exported"
    assert set(_lib._PROTOS) == set(declared), "ctypes prototypes out of sync with the header"
    assert L.l2hmc_abi_version() == 1


def test_library_exports_nothing_the_header_does_not_declare(L):
    """The shipped library carries no test / debug hooks: every exported C symbol `l2hmc_*` is declared in
    include/l2hmc_hip.h (run-time switches are plan flags or plan fields), and the library reads no environment
    variable."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("l2hmc_")}
    assert exported == set(_lib.declared_symbols()), sorted(exported ^ set(_lib.declared_symbols()))
    assert not [s for s in exported if "debug" in s]
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "getenv" not in und


def test_torch_ops_are_registered_and_have_no_cpu_implementation():
    """the stateless operators are also visible as torch.ops.l2hmc.* (l2hmc_amd/torch_ops.py, thin
    dispatchers onto the C ABI); a CPU tensor raises -- there is no CPU path behind them either."""
    import l2hmc_amd.torch_ops as T
    for name in T.OPS:
        assert hasattr(torch.ops.l2hmc, name), name
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.l2hmc.kinetic_energy(torch.zeros(3, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.l2hmc.u1_action_force(torch.zeros(2, 128), 8, 8, 2.0)


def test_struct_layouts_match_the_c_header(tmp_path):
    """sizeof/offsetof as gcc sees include/l2hmc_hip.h vs the ctypes mirrors: a mismatch would corrupt every call."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "l2hmc_hip.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(l2hmc_dense_net),'
        ' sizeof(l2hmc_conv3d_front), sizeof(l2hmc_gauge_plan), sizeof(l2hmc_mog_target), sizeof(l2hmc_small_plan),'
        ' offsetof(l2hmc_dense_net, packed), offsetof(l2hmc_gauge_plan, masks), offsetof(l2hmc_gauge_plan, vfront),'
        ' offsetof(l2hmc_small_plan, target), sizeof(l2hmc_dense_grads), sizeof(l2hmc_conv3d_grads)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.dirname(_lib.HEADER_PATH), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(_lib.DenseNet), C.sizeof(_lib.Conv3DFront), C.sizeof(_lib.GaugePlan), C.sizeof(_lib.MogTarget),
            C.sizeof(_lib.SmallPlan), _lib.DenseNet.packed.offset, _lib.GaugePlan.masks.offset,
            _lib.GaugePlan.vfront.offset, _lib.SmallPlan.target.offset, C.sizeof(_lib.DenseGrads),
            C.sizeof(_lib.Conv3DGrads)]
    assert got == want


def test_empty_inputs_are_ok_without_a_gpu(L):
    assert L.l2hmc_u1_action_force(None, 0, 8, 8, 1.0, None, None, None, None, None) == 0
    assert L.l2hmc_kinetic_energy(None, 0, 128, None, None) == 0
    assert L.l2hmc_fill_normal(None, 0, 1, 0, None) == 0
    assert L.l2hmc_accept_prob(None, None, None, 0, None, None) == 0


def test_bad_arguments_are_rejected_on_the_host(L):
    assert L.l2hmc_u1_action_force(None, 4, 8, 8, 1.0, None, None, None, None, None) == 1
    assert b"NULL" in L.l2hmc_last_error()
    assert L.l2hmc_u1_action_force(None, -1, 8, 8, 1.0, None, None, None, None, None) == 1
    assert L.l2hmc_lf_update_v(None, None, None, None, None, 0.1, 2, 4, 128, None, None, None) == 1
    with pytest.raises(ValueError):
        _lib.check(L.l2hmc_kinetic_energy(None, 3, 0, None, None))
    # any positive widths are accepted by the layered kernels (odd ones take the bounds-checked instantiation);
    # non-positive ones are refused, and a shape without a whole-trajectory kernel says so through pack_bytes
    net = _lib.DenseNet(D=2, H=0, Ka=2, Kb=2)
    assert L.l2hmc_stq_dense(C.byref(net), None, None, None, 1.0, 0.0, 4, None, None, None, None, 0, None) == 1
    assert b"must be positive" in L.l2hmc_last_error()
    net = _lib.DenseNet(D=72, H=288, Ka=72, Kb=72)
    assert L.l2hmc_stq_dense(C.byref(net), None, None, None, 1.0, 0.0, 4, None, None, None, None, 0, None) == 1
    assert b"NULL" in L.l2hmc_last_error()                    # the shape itself passed the check
    assert L.l2hmc_dense_pack_bytes(C.byref(net)) == 0
    # training keeps the multiple-of-32 requirement
    tplan = _lib.GaugePlan(T=6, X=6, num_steps=2, hmc=0, xnet=net, vnet=net, masks=16)   # any non-NULL address: host check only
    assert L.l2hmc_gauge_train_forward(C.byref(tplan), 1.0, None, None, None, 4, None, None, None, None, None, 0,
                                       None) == 1
    assert b"multiples of 32" in L.l2hmc_last_error()
    plan = _lib.GaugePlan(T=8, X=8, num_steps=0, hmc=1)
    assert L.l2hmc_gauge_trajectory(C.byref(plan), 1.0, None, None, None, 4, None, None, None, None, None, 0,
                                    None) == 1


def test_training_entry_points_check_arguments_on_the_host(L):
    """The training ABI (include/l2hmc_hip.h, training section): refused before any launch, no GPU needed."""
    net = _lib.DenseNet(D=128, H=512, Ka=128, Kb=128)
    hmc = _lib.GaugePlan(T=8, X=8, num_steps=10, hmc=1, xnet=net, vnet=net)
    assert L.l2hmc_gauge_train_ws_bytes(C.byref(hmc), 64) == 0                     # nothing to train
    assert L.l2hmc_gauge_train_forward(C.byref(hmc), 1.0, None, None, None, 4, None, None, None, None, None, 0,
                                       None) == 1
    plan = _lib.GaugePlan(T=8, X=8, num_steps=10, hmc=0, xnet=net, vnet=net)      # weights are NULL
    assert L.l2hmc_gauge_train_forward(C.byref(plan), 1.0, None, None, None, 4, None, None, None, None, None, 0,
                                       None) == 1
    assert L.l2hmc_gauge_train_backward(C.byref(plan), 1.0, None, 0, None, None, None, None, None, None, None, None,
                                        None, 0, None) == 1
    small = _lib.DenseNet(D=32, H=96, Ka=32, Kb=32)
    bad = _lib.GaugePlan(T=4, X=4, num_steps=2, hmc=0, xnet=small, vnet=net)        # vnet widths do not match D
    assert L.l2hmc_gauge_train_forward(C.byref(bad), 1.0, None, None, None, 4, None, None, None, None, None, 0,
                                       None) == 1
    assert L.l2hmc_gauge_loss_backward(8, 8, 2.0, None, None, None, None, 4, 7, 1., 1., 1., 1., 1., None, None, None,
                                       None, None) == 1                               # metric code out of range
    assert L.l2hmc_gauge_loss_backward(8, 8, 2.0, None, None, None, None, 0, 4, 1., 1., 1., 1., 1., None, None, None,
                                       None, None) == 0                               # empty batch
    assert L.l2hmc_adam_step(None, None, None, None, -1, 1e-3, .9, .999, 1e-8, None, 0., 0, 0, None) == 1
    assert L.l2hmc_adam_step(None, None, None, None, 0, 1e-3, .9, .999, 1e-8, None, 0., 0, 0, None) == 0
    assert L.l2hmc_grad_sumsq(None, 5, 0, 0, None, 0, None) == 1
    sp = _lib.SmallPlan(x_dim=2, num_nodes=50, trajectory_length=10, hmc=1)
    assert L.l2hmc_small_train_ws_bytes(C.byref(sp), 64) == 0
    assert L.l2hmc_small_train_step(C.byref(sp), None, None, None, 8, 0.1, 1., None, None, None, None, None, None, 0,
                                    None) == 1
    # training tape of the benchmark shape: 2 x 2048 chains, 10 LF -> about 2.2 GB
    plan_ws = L.l2hmc_gauge_train_ws_bytes(C.byref(plan), 4096)
    assert 1.5e9 < plan_ws < 3e9


def test_workspace_queries(L):
    net = _lib.DenseNet(D=128, H=512, Ka=128, Kb=128)
    # the 16-row image (every weight once) + the sub-tile form's image (heads padded to two 64-lane blocks per wave)
    assert L.l2hmc_dense_pack_bytes(C.byref(net)) == 4 * ((256 * 512 + 512 * 512 + 384 * 512) + (256 * 512 + 512 * 512 + 512 * 512))
    plan = _lib.GaugePlan(T=8, X=8, num_steps=10, hmc=0, xnet=net, vnet=net)
    one = L.l2hmc_gauge_ws_bytes(C.byref(plan), 4096)
    assert one >= 2 * 4096 * 512 * 4 + 4096 * 128 * 4
    both = L.l2hmc_gauge_transition_ws_bytes(C.byref(plan), 2048, 1)
    assert both >= one + 2 * 4096 * 128 * 4
    assert L.l2hmc_stq_ws_bytes(100, 512) >= 2 * 100 * 512 * 4


def test_product_path_has_no_cpu_fallback():
    """Host classes refuse CPU tensors instead of computing on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("checked on the GPU-less builder")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.dev_ptr(torch.zeros(4), name="x")
    from l2hmc_amd.network import GenericNet
    net = GenericNet(model_name='XNet', device=torch.device("cpu"), x_dim=128, num_hidden=512, factor=2.,
                     name_scope='position', links_shape=(8, 8, 2))
    with pytest.raises(RuntimeError):
        net([torch.zeros(4, 128), torch.zeros(4, 128), torch.tensor([[1., 0.]])])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(tmp_path, "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_step_launch_plan_covers_every_batch(L):
    """launch_fused_step (csrc/fused_traj.hip) cuts the chain-rows of one MCMC step into at most three launches:
    whole rounds of 32-row workgroups, one 16-row round, a sub-tile launch.  Host logic only: every plan covers its
    rows exactly once, cuts at even row counts (both directions of a chain stay in one workgroup), uses the
    sub-tile form only where it has a workgroup per CU or fewer, and never needs more rounds than 16-row tiles
    alone would."""
    fn = L.l2hmc_gauge_step_plan
    rows_out, rpw_out = (C.c_int64 * 3)(), (C.c_int * 3)()
    cus = 256
    t16, t32 = 1.0, 1.8                                   # one round of 16-row / 32-row workgroups (measured 1.59 / 2.86 ms)

    def cost(rows, rpw):
        if rpw == 16:
            return t16 * -(-rows // (16 * cus))
        if rpw == 32:
            return t32 * -(-rows // (32 * cus))
        return {4: 0.58, 8: 0.68, 12: 0.93}[rpw]          # sub-tile launches (0.91 / 1.07 / 1.46 ms)
    for rows in list(range(2, 40000, 2)) + [1, 3, 4097, 8193, 1 << 20, (1 << 20) + 2]:
        n = fn(rows, cus, rows_out, rpw_out)
        parts = [(rows_out[i], rpw_out[i]) for i in range(n)]
        assert 1 <= n <= 3 and sum(r for r, _ in parts) == rows, (rows, parts)
        assert all(r > 0 and w in (4, 8, 12, 16, 32) for r, w in parts), (rows, parts)
        assert all(r % 2 == 0 for r, _ in parts[:-1]), (rows, parts)
        for r, w in parts:
            if w < 16:
                assert r <= w * cus, (rows, parts)
            if w == 32:
                assert r > 16 * cus, (rows, parts)
        assert sum(cost(r, w) for r, w in parts) <= t16 * -(-rows // (16 * cus)) + 1e-9, (rows, parts)
    assert [(rows_out[i], rpw_out[i]) for i in range(fn(4096, cus, rows_out, rpw_out))] == [(4096, 16)]
    assert [(rows_out[i], rpw_out[i]) for i in range(fn(8192, cus, rows_out, rpw_out))] == [(8192, 32)]
    assert [(rows_out[i], rpw_out[i]) for i in range(fn(12400, cus, rows_out, rpw_out))] == [(8192, 32), (4096, 16), (112, 4)]
