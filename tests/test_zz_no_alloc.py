"""The allocation-free entry point t2_spec_loss_noalloc (include/t2mi.h) against t2_spec_loss through the C-ABI: two calls
in a row, the second with a smaller input (the scratch array keeps the first call's partials)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call(ctx, fn, y_hat, y, mask):
    import torch
    from tacotron2_amd import _lib
    B, T, M = y_hat.shape
    out = torch.full((3,), float("nan"), dtype=torch.float32, device=ctx.device)
    with ctx.ordered():
        _lib.check(fn(ctx.handle, C.c_void_p(y_hat.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(mask.data_ptr()), B, T, M,
                      C.c_void_p(out.data_ptr())))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_no_alloc_spec_loss_matches_the_allocating_entry_point():
    import torch
    from oracle import t2_oracle as O
    from tacotron2_amd import _lib
    from tacotron2_amd.engine import shared_context
    ctx = shared_context(0)
    lib = ctx.lib
    rng = np.random.default_rng(7)
    for (B, T, M) in ((32, 800, 80), (3, 41, 80)):          # second call: smaller input, fewer partial blocks
        y_hat = rng.normal(size=(B, T, M)).astype(np.float32)
        y = rng.uniform(0, 1, (B, T, M)).astype(np.float32)
        lens = rng.integers(T // 2, T + 1, B)
        mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)
        d = [torch.tensor(a, device=ctx.device) for a in (y_hat, y, mask)]
        ref = _call(ctx, lib.t2_spec_loss, *d)
        got = _call(ctx, lib.t2_spec_loss_noalloc, *d)
        np.testing.assert_array_equal(got, ref)                                  # same kernels, same order: bit-exact
        np.testing.assert_allclose(got[2], O.spec_loss(y_hat, y, mask), rtol=2e-5)
    # sizes beyond the stated maximum are rejected
    big = torch.zeros((1, 1, 4), device=ctx.device)
    rc = lib.t2_spec_loss_noalloc(ctx.handle, C.c_void_p(big.data_ptr()), C.c_void_p(big.data_ptr()), C.c_void_p(big.data_ptr()),
                                  1 << 15, 1 << 14, 4, C.c_void_p(big.data_ptr()))
    assert rc == _lib.T2_EINVAL and b"exceed the maximum" in lib.t2_last_error()
