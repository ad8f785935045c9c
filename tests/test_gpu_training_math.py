"""GPU parity of the training-step math through the C-ABI against the
NumPy oracle.  Tolerances: the losses are means of O(1) fp32 terms accumulated in another order than
NumPy's (1e-5 relative); one Adam step is element-wise fp32 arithmetic (ulp-level: 1e-6 absolute on
parameters of O(1), 1e-5 relative / 1e-8 absolute on m, 1e-5 relative on v)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import t2_oracle as O


@pytest.fixture(scope="module")
def ctx():
    from tacotron2_amd.engine import shared_context
    return shared_context(0)


@pytest.mark.parametrize("B,T,M", [(3, 7, 5), (2, 40, 80), (32, 800, 80), (1, 1, 1)])
def test_spec_loss(ctx, B, T, M):
    from tacotron2_amd import training as TR
    rng = np.random.default_rng(B * 1000 + T)
    y_hat = rng.standard_normal((B, T, M)).astype(np.float32)
    y = rng.uniform(0, 1, (B, T, M)).astype(np.float32)
    lens = rng.integers(1, T + 1, size=B)
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)
    parts = TR.spec_loss_parts(ctx, y_hat, y, mask).cpu().numpy()
    num, cnt = O.loss_parts(np.abs(y_hat - y), mask[:, :, None])
    assert parts[1] == cnt
    np.testing.assert_allclose(parts[0], num, rtol=1e-5)
    np.testing.assert_allclose(parts[2], O.spec_loss(y_hat, y, mask), rtol=1e-5)
    np.testing.assert_allclose(float(TR.spec_loss(ctx, y_hat, y, mask)), O.spec_loss(y_hat, y, mask), rtol=1e-5)


def test_spec_loss_all_masked_and_weighted(ctx):
    from tacotron2_amd import training as TR
    rng = np.random.default_rng(5)
    y_hat, y = rng.standard_normal((2, 6, 8)).astype(np.float32), rng.standard_normal((2, 6, 8)).astype(np.float32)
    assert float(TR.spec_loss(ctx, y_hat, y, np.zeros((2, 6), np.float32))) == 0.0        # _safe_div
    mask = rng.uniform(0.5, 2.0, (2, 6)).astype(np.float32)                                # non-binary weights
    np.testing.assert_allclose(float(TR.spec_loss(ctx, y_hat, y, mask)), O.spec_loss(y_hat, y, mask), rtol=1e-5)
    with pytest.raises(ValueError):
        TR.spec_loss(ctx, y_hat, y, np.ones((2, 5), np.float32))
    with pytest.raises(NotImplementedError):
        TR.spec_loss(ctx, y_hat, y, mask, n_priority_freq=4, priority_w=0.5)


@pytest.mark.parametrize("B,S", [(3, 6), (32, 400), (1, 1)])
def test_binary_loss(ctx, B, S):
    from tacotron2_amd import training as TR
    rng = np.random.default_rng(B + S)
    x = (rng.standard_normal((B, S, 1)) * 8).astype(np.float32)          # includes |x| large: the stable form matters
    lens = rng.integers(1, S + 1, size=B)
    done = (np.arange(S)[None, :] >= (lens[:, None] - 1)).astype(np.float32)     # zeros(S-1) || 1, padded with 1
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.float32)
    got = float(TR.binary_loss(ctx, x, done, mask))
    np.testing.assert_allclose(got, O.binary_loss(x, done, mask), rtol=1e-5, atol=1e-7)
    parts = TR.binary_loss_parts(ctx, x[..., 0], done, mask).cpu().numpy()
    assert parts[1] == np.count_nonzero(mask)


@pytest.mark.parametrize("n", [8, 1003, 1 << 20])
def test_adam_with_global_norm_clipping(ctx, n):
    import torch
    from tacotron2_amd import training as TR
    rng = np.random.default_rng(n)
    p0 = rng.standard_normal(n).astype(np.float32)
    opt = TR.AdamOptimizer(ctx, n, learning_rate=lambda step: TR.learning_rate_decay(0.002, step), beta1=0.9, beta2=0.999, epsilon=1e-8)
    p = torch.tensor(p0, device="cuda")
    po, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    step = 0
    for it in range(4):
        scale = [3.0, 0.01, 1.0, 0.3][it]                    # norms above and below the clip threshold
        g = (rng.standard_normal(n) * scale / np.sqrt(n)).astype(np.float32)
        gt = torch.tensor(g, device="cuda")
        norm = float(opt.global_norm(gt))
        np.testing.assert_allclose(norm, O.global_norm([g]), rtol=1e-5)
        (gc,), _ = O.clip_by_global_norm([g], 1.0)
        po, m, v = O.adam_apply(po, gc, m, v, it + 1, TR.learning_rate_decay(0.002, step))
        step = opt.apply_gradients(gt, p, global_step=step, clip_norm=1.0)
        np.testing.assert_allclose(p.cpu().numpy(), po, atol=1e-6, rtol=0)
        gm, gv, gstep = opt.state()
        assert gstep == it + 1
        np.testing.assert_allclose(gm.cpu().numpy(), m, rtol=1e-5, atol=1e-8)       # m cancels near zero: absolute bound
        np.testing.assert_allclose(gv.cpu().numpy(), v, rtol=1e-5, atol=1e-14)
    assert step == 4
    opt.close()


def test_adam_grad_scale_equals_mean_of_ranks_and_state_roundtrip(ctx):
    import torch
    from tacotron2_amd import training as TR
    n, world = 4096, 4
    rng = np.random.default_rng(11)
    gs = [(rng.standard_normal(n) * 0.05).astype(np.float32) for _ in range(world)]
    gsum = np.sum(gs, axis=0, dtype=np.float32)
    p0 = rng.standard_normal(n).astype(np.float32)
    a = TR.AdamOptimizer(ctx, n, learning_rate=1e-3)
    b = TR.AdamOptimizer(ctx, n, learning_rate=1e-3)
    pa, pb = torch.tensor(p0, device="cuda"), torch.tensor(p0, device="cuda")
    a.apply_gradients(torch.tensor(gsum, device="cuda"), pa, clip_norm=1.0, grad_scale=1.0 / world)
    gmean = (gsum * np.float32(1.0 / world)).astype(np.float32)
    b.apply_gradients(torch.tensor(gmean, device="cuda"), pb, clip_norm=1.0)
    np.testing.assert_allclose(pa.cpu().numpy(), pb.cpu().numpy(), atol=1e-7)
    # resume: a fresh optimizer loaded with a's slots continues identically
    c = TR.AdamOptimizer(ctx, n, learning_rate=1e-3)
    c.load_state(*a.state())
    pc = pa.clone()
    g2 = torch.tensor((rng.standard_normal(n) * 0.05).astype(np.float32), device="cuda")
    a.apply_gradients(g2, pa)
    c.apply_gradients(g2, pc)
    assert torch.equal(pa, pc)
    with pytest.raises(ValueError):
        a.apply_gradients(g2[:-1], pa)
    for o in (a, b, c):
        o.close()


def test_optimizer_rejects_bad_arguments(ctx):
    from tacotron2_amd import training as TR
    from tacotron2_amd._lib import T2Error
    with pytest.raises(T2Error):
        TR.AdamOptimizer(ctx, 0)
    with pytest.raises(T2Error):
        TR.AdamOptimizer(ctx, 16, beta1=1.0)
