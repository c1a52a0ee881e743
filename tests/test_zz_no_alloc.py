"""nerfhip_render_dev_static -- the one entry point that allocates nothing (intermediates in a static device array,
include/nerfhip.h) -- against nerfhip_render_dev on the same inputs: rgb bit for bit, on two calls in a row, the second
with a smaller input (the array keeps the first call's contents); n beyond the stated maximum is refused."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,dtype", [("small_dir", "f32"), ("full", "f32"), ("full", "bf16"), ("shipped", "f32"), ("full", "f16x3")])
def test_render_dev_static_equals_render_dev(name, dtype):
    import torch
    import gpu_util as G
    from nerf_amd import NerfHipError
    c = H.CONFIGS[name]
    n1, n2 = 3000, 1111
    ro, rd, b, _ = H.make_rays(n1, 23)
    params = H.make_params(c, 6)
    ref = G.make_block(c, max_rays=n1, dtype=dtype)
    blk = G.make_block(c, max_rays=8, dtype=dtype)           # its own workspace is far too small for these calls
    for x in (ref, blk):
        x.set_params(params)
    cap = blk.render_static_max_rays()
    assert cap >= n1
    if name == "full":
        assert cap == 11358                                   # the figure include/nerfhip.h states
    dro, drd, db = G.dev(ro), G.dev(rd), G.dev(b)
    for n in (n1, n2):                                        # second call smaller: stale contents must not matter
        want, got = G.empty(n, 3), torch.full((n, 3), -7.0, device="cuda")
        ref.render_dev(dro[:n], drd[:n], db[:n], drd[:n], n, want)
        blk.render_dev_static(dro[:n], drd[:n], db[:n], drd[:n], n, got)
        ref.synchronize(); blk.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())
    # the ordinary entry point of the small handle still works and still has its own bound
    out8 = G.empty(8, 3)
    blk.render_dev(dro[:8], drd[:8], db[:8], drd[:8], 8, out8)
    blk.synchronize()
    np.testing.assert_array_equal(out8.cpu().numpy(), want.cpu().numpy()[:8])
    with pytest.raises(NerfHipError, match="max_rays"):
        blk.render_dev(dro[:9], drd[:9], db[:9], drd[:9], 9, out8)
    with pytest.raises(NerfHipError, match="static scratch"):
        blk.render_dev_static(dro, drd, db, drd, cap + 1, want)
    ref.close(); blk.close()
