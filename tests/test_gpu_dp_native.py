"""The native data-parallel entry points of the C ABI (nerfhip_dp_*, csrc/dp_rccl.inc: RCCL bound with dlopen, no torch
in the exchange) -- the replacement of the multi-device branch of runNerf.scala:95-106.  The test box has one GPU and
RCCL wants one GPU per rank, so what runs here is a 1-rank communicator: rendezvous id, init, broadcast of the
state, the in-place all-reduce of the gradient buffer on the handle's stream, Adam with 1/world -- equal to the plain
train_step; plus the call-sequence errors.  N > 1 is the driver's 8-GPU run (bench.py --native-rccl)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import torch
    assert torch.cuda.is_available()
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16"])
def test_native_rccl_world1_equals_train_step(G, dtype):
    from nerf_amd import NerfBlock, NerfHipError
    c = H.CONFIGS["small_dir"]
    n = 128
    ro, rd, b, label = H.make_rays(n, 2)
    params = H.make_params(c, 1)
    a, ref = G.make_block(c, dtype=dtype), G.make_block(c, dtype=dtype)
    for blk in (a, ref):
        blk.set_params(params)
    assert a.dp_world() == (0, 1)
    with pytest.raises(NerfHipError):
        a.dp_allreduce_grads()                      # not initialised
    uid = NerfBlock.dp_unique_id()
    assert len(uid) == 128 and any(uid)
    a.dp_init(0, 1, uid)
    assert a.dp_world() == (0, 1)
    with pytest.raises(NerfHipError):
        a.dp_init(0, 1, uid)                        # twice
    with pytest.raises(NerfHipError):
        a.dp_allreduce_grads()                      # no gradient yet
    a.dp_broadcast_state(0)
    np.testing.assert_array_equal(a.get_params(), params)
    dev = [G.dev(x) for x in (ro, rd, b, rd, label)]
    loss = np.zeros(2, np.float32)
    for it in range(3):
        if it == 1:     # the three calls separately
            a.train_backward_dev(*dev, n, loss)
            g_before = a.get_grads()
            a.dp_allreduce_grads()
            np.testing.assert_array_equal(a.get_grads(), g_before)     # sum over one rank
            a.optimizer_step(1.0)
        else:
            a.dp_train_step_dev(*dev, n, loss)
        a.synchronize()
        lf, lc = ref.train_step(ro, rd, b, rd, label)
        assert loss[0] == pytest.approx(lf, rel=1e-6) and loss[1] == pytest.approx(lc, rel=1e-6)
    np.testing.assert_allclose(a.get_params(), ref.get_params(), rtol=0, atol=2e-4 if dtype == "bf16" else 2e-6)
    assert a.get_opt_state()[2] == 3
    a.set_profiling(True)
    a.dp_train_step_dev(*dev, n, None)
    prof = a.get_profile()
    assert prof["comm_ms"] > 0 and prof["mlp_wgrad_ms"] > 0
    a.dp_destroy()
    assert a.dp_world() == (0, 1)
    a.dp_destroy()                                  # idempotent
    a.close(); ref.close()


def test_native_rccl_single_thread_group_forms(G):
    """nerfhip_dp_init_all / nerfhip_dp_allreduce_grads_all (one host thread owning all devices, the reference's shape:
    one JVM, nerfConfig.device: Array[Device]) with the one device this box has; two handles on ONE device are refused."""
    import ctypes as C
    from nerf_amd import _lib
    c = H.CONFIGS["small_sh"]
    n = 64
    ro, rd, b, label = H.make_rays(n, 5)
    params = H.make_params(c, 2)
    a, other, ref = G.make_block(c), G.make_block(c), G.make_block(c)
    for blk in (a, ref):
        blk.set_params(params)
    lib = _lib.load()
    two = (C.c_void_p * 2)(a._h, other._h)
    assert lib.nerfhip_dp_init_all(two, 2) == -1                       # NERFHIP_E_INVALID: same device twice
    assert b"one device" in lib.nerfhip_last_error(other._h)
    one = (C.c_void_p * 1)(a._h)
    assert lib.nerfhip_dp_init_all(one, 1) == 0
    assert a.dp_world() == (0, 1)
    a.train_backward(ro, rd, b, rd, label)
    assert lib.nerfhip_dp_allreduce_grads_all(one, 1) == 0
    a.optimizer_step(1.0)
    ref.train_step(ro, rd, b, rd, label)
    np.testing.assert_allclose(a.get_params(), ref.get_params(), rtol=0, atol=2e-6)
    for blk in (a, other, ref):
        blk.close()                                                     # destroys the communicator too


def test_zero_grads_takes_part_in_a_step(G):
    """a replica with an empty shard (runNerf.scala:95 split of a short last batch): zero gradient, Adam still steps"""
    c = H.CONFIGS["small_dir"]
    blk = G.make_block(c)
    blk.set_params(H.make_params(c, 1))
    ro, rd, b, label = H.make_rays(32, 2)
    blk.train_step(ro, rd, b, rd, label)
    p1 = blk.get_params()
    blk.zero_grads()
    assert not blk.get_grads().any()
    blk.optimizer_step(0.5)
    m, v, step = blk.get_opt_state()
    assert step == 2
    p2 = blk.get_params()
    assert np.isfinite(p2).all() and (p2 != p1).any()      # momentum of the first step still moves the weights
    blk.close()
