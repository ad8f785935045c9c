"""GPU parity of the Tacotron-1 PostNet training step (BASELINE.json configs[4]; TacotronV1PostNetModel.model_fn(TRAIN),
tacotron/models.py:190-233) through the C-ABI (t2v1_postnet_train_*) against the float64 torch restatement
(oracle/v1_torch_train.py, gradients by autograd).  Tolerances: forward 1e-3 absolute (north_star); gradients
2e-3 * max|reference gradient of that variable| + 1e-9, with the scale asserted non-degenerate, plus -- per element -- the
oracle's own KINK SPREAD: the loss is piecewise smooth (ReLU after the batch norms and in the highway layers), and an fp32 forward
pass can round a ReLU input onto the other side of zero, after which the hand-written backward correctly differentiates the
neighbouring piece.  Measured: at (B, T) = (5, 23) one highway_3 input of 1.99e-6 (the smallest of the pass) flipped and moved
its column of highway_3/H/kernel by 4.6e-3 of the scale, every other column agreeing to 1e-7.  The spread is the change of the
oracle's gradient when every ReLU input within 1e-5 of zero is differentiated on its other side (relu_shift); it is zero for
most elements and the test asserts that it stays a small correction."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import t2_oracle as O
from oracle import v1_torch_train as VT
from tacotron2_amd.hparams import V1_PROFILE
from tacotron2_amd.weights_v1 import init_weights_v1

ATOL = 1e-3
KINK = 1e-5          # ReLU inputs this close to zero may be differentiated on either side (5x the flip that was observed)
# the PostNet of the reference's hparams (hparams.py:52-58) on a narrower spectrogram: 128 GRU units per direction as published
POST = V1_PROFILE.replace(num_freq=257)
POST_NARROW = V1_PROFILE.replace(num_freq=64, post_net_conv_channels=32, post_net_max_filter_width=5, post_net_projection1_out_channels=64,
                                 post_net_num_highway=2)
# projection2 == GRU width: no adjustment layer (modules.py:216)
POST_NOADJ = V1_PROFILE.replace(num_mels=128, num_freq=96, post_net_conv_channels=16, post_net_max_filter_width=3,
                                post_net_projection1_out_channels=32, post_net_projection2_out_channels=128, post_net_num_highway=1)


def _np(t):
    return t.detach().cpu().numpy()


def _case(d, B, T, seed, ragged=True):
    rng = np.random.default_rng(seed)
    mel = rng.uniform(0, 1, (B, T, d.num_mels)).astype(np.float32)
    spec = rng.uniform(0, 1, (B, T, d.num_freq)).astype(np.float32)
    lens = rng.integers(max(1, T // 2), T + 1, size=B) if ragged else np.full(B, T)
    lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)
    return mel, spec, mask


def _setup(d, seed=11):
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import EngineV1, PostNetTrainer
    w = init_weights_v1(d, seed, perturb=True)
    eng = EngineV1(d, w)
    pw = PostNetTrainer.postnet_weights(w)
    spec = TR.flatten_spec(pw)
    flat = torch.tensor(TR.flatten(pw, spec), device="cuda")
    return w, eng, pw, spec, flat


@pytest.mark.parametrize("d,B,T", [(POST_NARROW, 3, 11), (POST_NARROW, 17, 9), (POST_NOADJ, 2, 7), (POST, 5, 23), (POST, 33, 40)])
def test_postnet_training_forward_and_every_gradient_against_autograd(d, B, T):
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import PostNetTrainer
    w, eng, pw, spec, flat = _setup(d)
    mel, target, mask = _case(d, B, T, 100 * B + T)
    ref = VT.postnet_train_forward_backward(d, w, mel, target, mask)
    side = [VT.postnet_train_forward_backward(d, w, mel, target, mask, relu_shift=sh)["grads"] for sh in (KINK, -KINK)]
    tr = PostNetTrainer(eng, B, T, spec)
    out = tr.forward(flat, mel)
    np.testing.assert_allclose(_np(out), ref["spec"], atol=ATOL)
    grads = torch.full_like(flat, float("nan"))               # every trainable slice must be written
    parts = _np(tr.backward(target, mask, grads))
    assert abs(parts[2] - ref["loss"]) <= 2e-4 * max(1.0, abs(ref["loss"]))
    assert parts[1] == mask.sum() * d.num_freq
    got = TR.unflatten(grads, spec)
    checked = 0
    for name, g_ref in ref["grads"].items():
        if name.endswith(("moving_mean", "moving_variance")):
            continue
        scale = float(np.abs(g_ref).max())
        assert scale >= 1e-7, (name, scale)                   # a vanishing reference gradient would make the comparison vacuous
        assert np.isfinite(got[name]).all(), name
        spread = np.abs(side[0][name] - g_ref) + np.abs(side[1][name] - g_ref)
        assert float(np.median(spread)) <= 5e-3 * scale and float(spread.max()) <= 0.1 * scale, (name, "kink spread too wide to test")
        excess = np.abs(got[name] - g_ref) - (2e-3 * scale + 1e-9 + 1.5 * spread)
        assert float(excess.max()) <= 0.0, (name, float(np.abs(got[name] - g_ref).max()), scale, float(spread.max()))
        checked += 1
    assert checked == sum(1 for n in pw if not n.endswith(("moving_mean", "moving_variance")))
    # a second backward without a forward is refused (the tape is consumed)
    with pytest.raises(Exception):
        tr.backward(target, mask, grads)
    tr.close()
    eng.close()


def test_postnet_training_data_parallel_count_scales_the_gradient():
    """loss_count (the non-zero weight count of a larger data-parallel batch) replaces this batch's own count in the gradient."""
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import PostNetTrainer
    d, B, T = POST_NARROW, 4, 9
    w, eng, pw, spec, flat = _setup(d)
    mel, target, mask = _case(d, B, T, 5)
    tr = PostNetTrainer(eng, B, T, spec)
    g1, g2 = torch.zeros_like(flat), torch.zeros_like(flat)
    tr.forward(flat, mel)
    own = float(_np(tr.backward(target, mask, g1))[1])
    tr.forward(flat, mel)
    tr.backward(target, mask, g2, loss_count=torch.tensor([2.5 * own], device="cuda"))
    ref = VT.postnet_train_forward_backward(d, w, mel, target, mask, loss_count=2.5 * own)
    got = TR.unflatten(g2, spec)
    for name in ("v1/postnet/linear/kernel", "v1/postnet/cbhg/conv1d_K3/kernel", "v1/postnet/cbhg/bidirectional_rnn/bw/gru_cell/gates/kernel"):
        scale = float(np.abs(ref["grads"][name]).max())
        assert float(np.abs(got[name] - ref["grads"][name]).max()) <= 2e-3 * scale + 1e-9
    torch.testing.assert_close(g1, g2 * 2.5, rtol=1e-5, atol=1e-5 * float(g1.abs().max()))
    tr.close()
    eng.close()


def test_postnet_model_fn_train_loop_against_oracle_and_export():
    """model_fn(TRAIN) three times (models.py:204-234: forward, L1 loss, clip_by_global_norm(1.0), Adam at the Noam rate,
    UPDATE_OPS) against the oracle loop; then EVAL / PREDICT use the trained variables (device-side export = a fresh engine
    built from the trained weights)."""
    from tacotron2_amd.tacotron_v1 import EngineV1, ModeKeys, TacotronV1PostNetModel
    d, B, T = POST_NARROW, 6, 13
    w = init_weights_v1(d, 21, perturb=True)
    eng = EngineV1(d, w)
    model = TacotronV1PostNetModel(d, eng, weights=w)
    hp = {"initial_learning_rate": 0.002, "decay_learning_rate": True, "adam_beta1": 0.9, "adam_beta2": 0.999, "adam_eps": 1e-8}
    names = sorted(k for k in w if k.startswith("v1/postnet/"))
    trainable = [k for k in names if not k.endswith(("moving_mean", "moving_variance"))]
    w_ref = {k: w[k].copy() for k in names}
    m_ref = {k: np.zeros_like(w[k]) for k in trainable}
    v_ref = {k: np.zeros_like(w[k]) for k in trainable}
    bn = ["v1/postnet/cbhg/conv1d_K%d" % k for k in range(1, d.post_net_max_filter_width + 1)] + ["v1/postnet/cbhg/proj1", "v1/postnet/cbhg/proj2"]
    lrs = []
    for it in range(3):
        mel, target, mask = _case(d, B, T, 50 + it)
        full = dict(w)
        full.update(w_ref)
        ref = VT.postnet_train_forward_backward(d, full, mel, target, mask)
        res = model.model_fn({"mel": mel}, {"spec": target, "spec_loss_mask": mask}, ModeKeys.TRAIN, hp)
        assert res["global_step"] == it + 1
        assert abs(float(res["loss"]) - ref["loss"]) <= 2e-4 * max(1.0, abs(ref["loss"])), (it, float(res["loss"]), ref["loss"])
        np.testing.assert_allclose(_np(res["spec"]), ref["spec"], atol=ATOL)
        lr = O.learning_rate_decay(0.002, it)
        lrs.append(lr)
        clipped, _ = O.clip_by_global_norm([ref["grads"][k] for k in trainable], 1.0)
        for k, g in zip(trainable, clipped):
            w_ref[k], m_ref[k], v_ref[k] = O.adam_apply(w_ref[k], g, m_ref[k], v_ref[k], it + 1, lr)
        n = B * T
        for p in bn:                                             # UPDATE_OPS (App. A4: Bessel-corrected variance)
            mu, var = ref["bn_stats"][p]
            q = p + "/batch_normalization"
            w_ref[q + "/moving_mean"] = (0.99 * w_ref[q + "/moving_mean"] + 0.01 * mu).astype(np.float32)
            w_ref[q + "/moving_variance"] = (0.99 * w_ref[q + "/moving_variance"] + 0.01 * var * n / (n - 1)).astype(np.float32)
    got = model.trained_weights()
    for p in bn:
        for s in ("/batch_normalization/moving_mean", "/batch_normalization/moving_variance"):
            np.testing.assert_allclose(got[p + s], w_ref[p + s], rtol=1e-4, atol=1e-6)
    tot = sum(lrs)
    for k in ("v1/postnet/linear/kernel", "v1/postnet/cbhg/proj1/kernel", "v1/postnet/cbhg/conv1d_K2/kernel",
              "v1/postnet/cbhg/bidirectional_rnn/fw/gru_cell/gates/kernel", "v1/postnet/cbhg/highway_1/T/kernel"):
        assert float(np.abs(got[k] - w[k]).max()) <= 1.05 * tot
        assert float(np.abs(got[k] - w[k]).max()) >= 0.5 * lrs[0]              # Adam's first step moves by lr wherever g != 0
        np.testing.assert_allclose(got[k], w_ref[k], atol=0.25 * tot)
    # EVAL after TRAIN: the inference copies are rebuilt from the trained variables
    mel, target, mask = _case(d, B, T, 99)
    ev = model.model_fn({"mel": mel}, {"spec": target, "spec_loss_mask": mask}, ModeKeys.EVAL)
    full = dict(w)
    full.update(got)
    eng2 = EngineV1(d, full)
    np.testing.assert_allclose(_np(ev["spec"]), _np(eng2.postnet(mel)), atol=2e-6, rtol=2e-6)
    eng2.close()
    eng.close()


def test_postnet_trainer_argument_errors():
    import torch
    from tacotron2_amd.tacotron_v1 import EngineV1, PostNet, PostNetTrainer
    from tacotron2_amd import training as TR
    d = POST_NARROW
    w, eng, pw, spec, flat = _setup(d)
    tr = PostNetTrainer(eng, 2, 5, spec)
    with pytest.raises(ValueError):
        tr.forward(flat, np.zeros((2, 6, d.num_mels), np.float32))
    with pytest.raises(ValueError):
        tr.forward(flat[:-4], np.zeros((2, 5, d.num_mels), np.float32))
    with pytest.raises(Exception):                                               # backward before any forward
        tr.backward(np.zeros((2, 5, d.num_freq), np.float32), np.ones((2, 5), np.float32), torch.zeros_like(flat))
    with pytest.raises(ValueError):
        PostNet(True, d.num_freq, engine=eng)                                    # no parameter buffer
    tr.close()
    # a GRU width the whole-sequence kernels are not built for is refused, not mis-computed
    d2 = V1_PROFILE.replace(num_freq=16, post_net_cbhg_out_units=64, post_net_conv_channels=8, post_net_max_filter_width=2,
                            post_net_projection1_out_channels=16, post_net_num_highway=1)
    w2 = init_weights_v1(d2, 3)
    eng2 = EngineV1(d2, w2)
    with pytest.raises(Exception):
        PostNetTrainer(eng2, 2, 5, TR.flatten_spec(PostNetTrainer.postnet_weights(w2)))
    eng2.close()
    eng.close()


def _dp_rank(rank, world, port, q):
    """One rank of the data-parallel PostNet training step: its half of the rows, the product's train_step(group=WORLD)."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tacotron2_amd.tacotron_v1 import EngineV1, TacotronV1PostNetModel
        d, w, mel, target, mask = _dp_case()
        rows = slice(3 * rank, 3 * rank + 3)
        eng = EngineV1(d, w)
        model = TacotronV1PostNetModel(d, eng, weights=w)
        res = model.train_step({"mel": mel[rows]}, {"spec": target[rows], "spec_loss_mask": mask[rows]},
                               {"initial_learning_rate": 0.002, "decay_learning_rate": False}, group=dist.group.WORLD)
        torch.cuda.synchronize()
        q.put((rank, float(res["loss"]), _np(model.flat_grads), _np(model.flat_params)))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


def _dp_case():
    d = POST_NARROW
    w = init_weights_v1(d, 17, perturb=True)
    mel, target, mask = _case(d, 6, 12, 77)
    return d, w, mel, target, mask


def test_postnet_training_step_two_ranks_share_the_loss_normalisation():
    """Data parallel through the product's own path: two processes (gloo; both on this one GPU -- no kernel of
    this step spans workgroups, so they may share it) run TacotronV1PostNetModel.train_step(group=...) on three rows each.
    The ranks sum their non-zero weight counts first, each differentiates its numerator / the GLOBAL count, gradients are
    summed: the summed gradient equals the sum of the oracle's per-rank gradients with that count, the loss is sum of numerators
    / global count, and both ranks end with identical parameters (batch statistics stay per replica, the moving statistics
    are averaged)."""
    import socket
    import torch.multiprocessing as mp
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import PostNetTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=110) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, w, mel, target, mask = _dp_case()
    count = float((mask != 0).sum() * d.num_freq)
    refs = [VT.postnet_train_forward_backward(d, w, mel[3 * r:3 * r + 3], target[3 * r:3 * r + 3], mask[3 * r:3 * r + 3], loss_count=count)
            for r in range(2)]
    spec = TR.flatten_spec(PostNetTrainer.postnet_weights(w))
    g_sum = TR.flatten({k: refs[0]["grads"][k] + refs[1]["grads"][k] for k in refs[0]["grads"]}, spec)
    loss = refs[0]["loss"] + refs[1]["loss"]                      # each is its numerator / the global count
    np.testing.assert_array_equal(res[0][3], res[1][3])           # identical replicas after the step
    for r in res:
        assert abs(r[1] - loss) <= 2e-4 * max(1.0, abs(loss))
        for name, shape, off, size in spec:
            if name.endswith(("moving_mean", "moving_variance")):
                continue
            scale = float(np.abs(g_sum[off:off + size]).max())
            assert float(np.abs(r[2][off:off + size] - g_sum[off:off + size]).max()) <= 5e-3 * scale + 1e-9, name


# ---- EncoderV1 training step (the encoder half of SingleSpeakerTacotronV1Model.model_fn(TRAIN), models.py:40-47) ----
ENC = V1_PROFILE.replace(num_symbols=40)                                  # published encoder sizes: K = 16 bank, 128-unit GRUs, no adjustment layer
ENC_NARROW = V1_PROFILE.replace(num_symbols=30, embedding_dim=32, encoder_prenet_out_units=(32, 48), conv_channels=16, max_filter_width=5,
                                projection1_out_channels=24, projection2_out_channels=48, num_highway=2)     # adjustment layer 48 -> 128, odd bank


def _enc_setup(d, seed=13):
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import EncoderV1Trainer, EngineV1
    w = init_weights_v1(d, seed, perturb=True)
    eng = EngineV1(d, w)
    ew = EncoderV1Trainer.encoder_weights(w)
    spec = TR.flatten_spec(ew)
    flat = torch.tensor(TR.flatten(ew, spec), device="cuda")
    return w, eng, ew, spec, flat


@pytest.mark.parametrize("d,B,T,drop,with_lens", [(ENC_NARROW, 3, 11, 0.5, True), (ENC_NARROW, 18, 7, 0.0, False), (ENC, 4, 21, 0.5, True),
                                                  (ENC, 33, 16, 0.5, False)])
def test_encoder_v1_training_forward_and_every_gradient_against_autograd(d, B, T, drop, with_lens):
    """Embedding -> PreNets (dropout masks from the shared counter-based generator) -> CBHG with batch statistics and sequence
    lengths: the encoder output and the gradient of every encoder variable and of the embedding table for a random
    d loss / d memory, against fp64 autograd; same kink-aware tolerance as the PostNet test above."""
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import EncoderV1Trainer
    w, eng, ew, spec, flat = _enc_setup(d)
    rng = np.random.default_rng(10 * B + T)
    ids = rng.integers(0, d.num_symbols, (B, T))
    lens = None
    if with_lens:
        lens = rng.integers(max(1, T // 2), T + 1, size=B).astype(np.int64)
        lens[0] = T
    dmem = rng.standard_normal((B, T, d.cbhg_out_units)).astype(np.float32)
    if lens is not None:
        dmem *= (np.arange(T)[None, :] < lens[:, None])[:, :, None]      # nothing flows into positions past a row's length
    seed = 77
    ref = VT.encoder_train_forward_backward(d, w, ids, lens, dmem, drop, seed)
    side = [VT.encoder_train_forward_backward(d, w, ids, lens, dmem, drop, seed, relu_shift=sh)["grads"] for sh in (KINK, -KINK)]
    tr = EncoderV1Trainer(eng, B, T, spec)
    mem = tr.forward(flat, ids, lens, drop_rate=drop, seed=seed)
    np.testing.assert_allclose(_np(mem), ref["memory"], atol=ATOL)
    grads = torch.full_like(flat, float("nan"))
    demb = tr.backward(dmem, grads, want_input_grad=True)
    assert tuple(demb.shape) == (B, T, d.embedding_dim) and np.isfinite(_np(demb)).all()
    got = TR.unflatten(grads, spec)
    checked = 0
    for name, g_ref in ref["grads"].items():
        if name.endswith(("moving_mean", "moving_variance")):
            continue
        scale = float(np.abs(g_ref).max())
        assert scale >= 1e-7, (name, scale)
        assert np.isfinite(got[name]).all(), name
        spread = np.abs(side[0][name] - g_ref) + np.abs(side[1][name] - g_ref)
        assert float(np.median(spread)) <= 5e-3 * scale and float(spread.max()) <= 0.1 * scale, (name, "kink spread too wide to test")
        excess = np.abs(got[name] - g_ref) - (2e-3 * scale + 1e-9 + 1.5 * spread)
        assert float(excess.max()) <= 0.0, (name, float(np.abs(got[name] - g_ref).max()), scale, float(spread.max()))
        checked += 1
    assert checked == sum(1 for n in ew if not n.endswith(("moving_mean", "moving_variance")))
    # the table's gradient is the scatter-add of the input gradient
    tab = np.zeros_like(w["embedding"])
    np.add.at(tab, ids.reshape(-1), _np(demb).reshape(-1, d.embedding_dim))
    np.testing.assert_allclose(got["embedding"], tab, atol=1e-5 * float(np.abs(tab).max()) + 1e-9)
    tr.close()
    eng.close()


def test_encoder_v1_block_in_training_mode_and_export():
    """EncoderV1(is_training=True).call on an embedded input (the block on its own, tacotron_v1.py:46-49): dropout seed advancing
    per call, moving statistics updated from the batch statistics, and export: the engine's inference encoder then equals a
    fresh engine built from the buffer's variables."""
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import EncoderV1, EngineV1
    d, B, T = ENC_NARROW, 5, 9
    w, eng, ew, spec, flat = _enc_setup(d)
    rng = np.random.default_rng(3)
    ids = rng.integers(0, d.num_symbols, (B, T))
    emb = O.embedding_lookup(ids, w["embedding"])
    with pytest.raises(ValueError):
        EncoderV1(True, engine=eng, prenet_out_units=d.encoder_prenet_out_units)
    enc = EncoderV1(True, d.cbhg_out_units, d.conv_channels, d.max_filter_width, d.projection1_out_channels, d.projection2_out_channels,
                    d.num_highway, d.encoder_prenet_out_units, 0.5, engine=eng, flat_params=flat, spec=spec, seed=5)
    y0, y1 = enc(emb), enc(emb)
    assert float((y0 - y1).abs().max()) > 1e-3                               # another seed, other masks
    dm = np.ones((B, T, d.cbhg_out_units), np.float32)
    np.testing.assert_allclose(_np(y1), VT.encoder_train_forward_backward(d, w, ids, None, dm, 0.5, 6)["memory"], atol=ATOL)
    before = TR.unflatten(flat, spec)
    enc.trainer.update_moving_stats(flat)
    ref = VT.encoder_train_forward_backward(d, w, ids, None, dm, 0.5, 6)
    after = TR.unflatten(flat, spec)
    n = B * T
    for p_, (mu, var) in ref["bn_stats"].items():
        q = p_ + "/batch_normalization"
        np.testing.assert_allclose(after[q + "/moving_mean"], 0.99 * before[q + "/moving_mean"] + 0.01 * mu, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(after[q + "/moving_variance"], 0.99 * before[q + "/moving_variance"] + 0.01 * var * n / (n - 1), rtol=1e-4, atol=1e-6)
    flat += 0.01 * torch.randn_like(flat)                                    # "trained" variables
    flat.abs_()                                                              # keeps the moving variances positive
    enc.trainer.export(flat)
    full = dict(w)
    full.update(TR.unflatten(flat, spec))
    eng2 = EngineV1(d, full)
    lens = np.array([9, 4, 6, 9, 2], np.int64)
    np.testing.assert_allclose(_np(eng.encode(ids, lens)), _np(eng2.encode(ids, lens)), atol=2e-6, rtol=2e-6)
    from tacotron2_amd._lib import T2Error
    with pytest.raises(T2Error):
        eng.set_option("gruseq", 0)                                          # the launch-per-step GRU operands were not rebuilt
    eng2.close()
    eng.close()


# ---- DecoderV1 training step ----
def _dec_case(d, B, T, S, seed):
    rng = np.random.default_rng(seed)
    r = d.outputs_per_step
    mem = (rng.standard_normal((B, T, d.cbhg_out_units)) * 0.5).astype(np.float32)
    lens = rng.integers(max(1, T // 2), T + 1, size=B).astype(np.int64)
    lens[0] = T
    tl = rng.integers(max(1, S * r // 2), S * r + 1, size=B)
    tl[0] = S * r
    sl = (tl + r - 1) // r
    labels = {"mel": rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32),
              "done": (np.arange(S)[None, :] >= (sl[:, None] - 1)).astype(np.float32),
              "spec_loss_mask": (np.arange(S * r)[None, :] < tl[:, None]).astype(np.float32),
              "binary_loss_mask": (np.arange(S)[None, :] < sl[:, None]).astype(np.float32)}
    return mem, lens, labels


@pytest.mark.parametrize("name,B,T,S,drop", [("tiny", 3, 7, 5, 0.5), ("mid", 5, 19, 6, 0.5), ("ref", 4, 23, 5, 0.5), ("ref", 33, 12, 4, 0.0)])
def test_decoder_v1_training_forward_and_every_gradient_against_autograd(name, B, T, S, drop):
    """Teacher-forced training forward of DecoderV1 (PreNet dropout from the shared counter-based generator) and the BPTT of
    spec_loss + binary_loss: mel, stop, alignments, losses, every v1/decoder/ and v1/attention/ gradient and d loss / d memory
    against fp64 autograd (no kinks apart from the PreNets' relu, whose inputs are targets through one or two layers)."""
    import torch
    from tests.test_gpu_v1 import DIMS
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import DecoderV1Trainer, EngineV1
    d = DIMS[name]
    w = init_weights_v1(d, 9, perturb=True)
    eng = EngineV1(d, w)
    dw = DecoderV1Trainer.decoder_weights(w)
    spec = TR.flatten_spec(dw)
    flat = torch.tensor(TR.flatten(dw, spec), device="cuda")
    mem, lens, labels = _dec_case(d, B, T, S, 1000 * B + T)
    ref = VT.decoder_train_forward_backward(d, w, mem, lens, labels, drop, 31)
    tr = DecoderV1Trainer(eng, B, T, S, spec)
    mel, stop, align = tr.forward(flat, mem, lens, labels["mel"], drop_rate=drop, seed=31)
    np.testing.assert_allclose(_np(mel), ref["mel"], atol=ATOL)
    np.testing.assert_allclose(_np(stop)[..., 0], ref["stop"], atol=ATOL)
    np.testing.assert_allclose(_np(align), ref["alignments"], atol=ATOL)
    grads = torch.full_like(flat, float("nan"))
    dmem, losses = tr.backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], grads)
    ls = _np(losses)
    assert abs(ls[0] - ref["mel_loss"]) <= 2e-4 * max(1.0, ref["mel_loss"]) and abs(ls[1] - ref["done_loss"]) <= 2e-4 * max(1.0, ref["done_loss"])
    assert abs(ls[2] - ref["loss"]) <= 2e-4 * max(1.0, ref["loss"])
    got = TR.unflatten(grads, spec)
    assert set(got) == set(ref["grads"])
    for k, g_ref in ref["grads"].items():
        scale = float(np.abs(g_ref).max())
        assert scale >= 1e-7, (k, scale)
        assert np.isfinite(got[k]).all(), k
        err = float(np.abs(got[k] - g_ref).max())
        assert err <= 2e-3 * scale + 1e-9, (k, err, scale)
    scale = float(np.abs(ref["dmemory"]).max())
    assert float(np.abs(_np(dmem) - ref["dmemory"]).max()) <= 2e-3 * scale + 1e-9
    for b_ in range(B):
        assert not _np(dmem)[b_, lens[b_]:].any()                            # values = memory * mask
    with pytest.raises(Exception):
        tr.backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], grads)
    tr.close()
    eng.close()


def test_decoder_v1_persistent_recurrences_equal_the_launch_per_stage_loops_and_fall_back_on_a_timeout():
    """Both recurrences of the decoder trainer as one persistent launch each (csrc/v1_dec_train.h, two chains per launch) against
    the same stages launched one by one (hipGraph loops): same device code, another summation order of the K split (16 waves
    instead of 8).  A hop timeout (injected) in either launch re-runs that recurrence launch by launch: same numbers as that path."""
    import torch
    from tests.test_gpu_v1 import DIMS
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import DecoderV1Trainer, EngineV1
    d = DIMS["ref"]
    B, T, S, drop = 20, 37, 7, 0.5
    w = init_weights_v1(d, 9, perturb=True)
    eng = EngineV1(d, w)
    dw = DecoderV1Trainer.decoder_weights(w)
    spec = TR.flatten_spec(dw)
    flat = torch.tensor(TR.flatten(dw, spec), device="cuda")
    mem, lens, labels = _dec_case(d, B, T, S, 77)
    tr = DecoderV1Trainer(eng, B, T, S, spec)

    def run(inject_fwd=0, inject_bwd=0):
        if inject_fwd:
            eng.set_option("inject_timeout", inject_fwd)
        mel, stop, align = tr.forward(flat, mem, lens, labels["mel"], drop_rate=drop, seed=5)
        if inject_bwd:
            eng.set_option("inject_timeout", inject_bwd)
        grads = torch.zeros_like(flat)                                       # (alignment padding between variables is never written)
        dmem, losses = tr.backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], grads)
        return [_np(x).copy() for x in (mel, stop, align, grads, dmem, losses)]

    n0 = eng.get_stat("train_persist_launches")
    pers = run()
    assert eng.get_stat("train_persist_launches") == n0 + 2 and eng.get_stat("persist_timeouts") == 0
    eng.set_option("train_persist", 0)
    loop = run()
    assert eng.get_stat("train_persist_launches") == n0 + 2
    for a, b_ in zip(pers, loop):
        scale = float(np.abs(b_).max())
        assert np.isfinite(a).all() and float(np.abs(a - b_).max()) <= 2e-5 * scale + 1e-9
    # a timed-out forward launch: the forward loop runs again launch by launch, and the next 16 calls stay on that schedule
    eng.set_option("train_persist", 1)
    got = run(inject_fwd=1)
    assert eng.get_stat("persist_timeouts") == 1 and eng.get_stat("train_persist_launches") == n0 + 3
    for a, b_ in zip(got, loop):
        np.testing.assert_array_equal(a, b_)
    # a timed-out backward launch after a good forward launch: accumulators start again, same numbers as the loops
    eng.set_option("train_persist", 1)
    got = run(inject_bwd=1)
    assert eng.get_stat("persist_timeouts") == 2 and eng.get_stat("train_persist_launches") == n0 + 5
    for i, (a, b_) in enumerate(zip(got, loop)):
        if i < 3:                                                            # forward: persistent launch
            scale = float(np.abs(b_).max())
            assert float(np.abs(a - b_).max()) <= 2e-5 * scale + 1e-9
        else:                                                                # backward: loops on the persistent forward's tape
            scale = float(np.abs(b_).max())
            assert np.isfinite(a).all() and float(np.abs(a - b_).max()) <= 2e-5 * scale + 1e-9
    tr.close()
    eng.close()


def test_single_speaker_v1_model_fn_train_loop_against_oracle():
    """SingleSpeakerTacotronV1Model.model_fn(TRAIN) (models.py:40-76) twice against the oracle loop (encoder oracle o decoder
    oracle through d memory, clip_by_global_norm(1.0), TF-form Adam at the Noam rate, UPDATE_OPS), then PREDICT with the trained
    variables = a fresh model built from them."""
    from tests.test_gpu_v1 import DIMS
    from tacotron2_amd.tacotron_v1 import ModeKeys, SingleSpeakerTacotronV1Model
    d = DIMS["ref"].replace(num_symbols=40, max_iters=6, min_iters=2)
    w = init_weights_v1(d, 19, perturb=True)
    model = SingleSpeakerTacotronV1Model(d, weights=w, seed=100)
    hp = {"initial_learning_rate": 0.002, "decay_learning_rate": True, "encoder_prenet_drop_rate": 0.5, "decoder_prenet_drop_rate": 0.5}
    B, T, S = 5, 11, 4
    names = sorted(k for k in w if not k.startswith("v1/postnet/"))
    trainable = [k for k in names if not k.endswith(("moving_mean", "moving_variance"))]
    w_ref = {k: w[k].copy() for k in w}
    m_ref = {k: np.zeros_like(w[k]) for k in trainable}
    v_ref = {k: np.zeros_like(w[k]) for k in trainable}
    lrs = []
    for it in range(2):
        rng = np.random.default_rng(40 + it)
        ids = rng.integers(0, d.num_symbols, (B, T))
        _, lens, labels = _dec_case(d, B, T, S, 70 + it)
        seed = 100 + it                                                      # model seed + global_step
        enc = VT.encoder_train_forward_backward(d, w_ref, ids, None, np.zeros((B, T, d.cbhg_out_units), np.float32), 0.5, seed)
        dec = VT.decoder_train_forward_backward(d, w_ref, enc["memory"], lens, labels, 0.5, seed)
        enc = VT.encoder_train_forward_backward(d, w_ref, ids, None, dec["dmemory"], 0.5, seed)
        res = model.model_fn({"source": ids, "source_length": lens}, labels, ModeKeys.TRAIN, hp)
        assert res["global_step"] == it + 1
        assert abs(float(res["loss"]) - dec["loss"]) <= 5e-4 * max(1.0, dec["loss"]), (it, float(res["loss"]), dec["loss"])
        np.testing.assert_allclose(_np(res["mel"]), dec["mel"], atol=2e-3)
        grads = dict(dec["grads"])
        grads.update(enc["grads"])
        lr = O.learning_rate_decay(0.002, it)
        lrs.append(lr)
        assert abs(res["learning_rate"] - lr) < 1e-12
        clipped, _ = O.clip_by_global_norm([grads[k] for k in trainable], 1.0)
        for k, g in zip(trainable, clipped):
            w_ref[k], m_ref[k], v_ref[k] = O.adam_apply(w_ref[k], g, m_ref[k], v_ref[k], it + 1, lr)
        n = B * T
        for p_, (mu, var) in enc["bn_stats"].items():
            q = p_ + "/batch_normalization"
            w_ref[q + "/moving_mean"] = (0.99 * w_ref[q + "/moving_mean"] + 0.01 * mu).astype(np.float32)
            w_ref[q + "/moving_variance"] = (0.99 * w_ref[q + "/moving_variance"] + 0.01 * var * n / (n - 1)).astype(np.float32)
    got = model.trained_weights()
    tot = sum(lrs)
    for k in ("v1/encoder/cbhg/proj1/batch_normalization/moving_mean", "v1/encoder/cbhg/conv1d_K3/batch_normalization/moving_variance"):
        np.testing.assert_allclose(got[k], w_ref[k], rtol=1e-3, atol=1e-5)
    for k in ("embedding", "v1/encoder/cbhg/proj1/kernel", "v1/decoder/attention_gru/gates/kernel", "v1/attention/query_layer/kernel",
              "v1/decoder/gru_2/candidate/kernel", "v1/decoder/out_projection/kernel"):
        assert float(np.abs(got[k] - w[k]).max()) <= 1.05 * tot
        assert float(np.abs(got[k] - w[k]).max()) >= 0.5 * lrs[0]
        np.testing.assert_allclose(got[k], w_ref[k], atol=0.3 * tot)
    np.testing.assert_array_equal(got["v1/postnet/linear/kernel"], w["v1/postnet/linear/kernel"])      # not a variable of this model
    # PREDICT after TRAIN uses the trained variables
    ids = np.random.default_rng(5).integers(0, d.num_symbols, (2, 9))
    feats = {"source": ids, "source_length": np.array([9, 6])}
    pred = model.predict(feats)
    fresh = SingleSpeakerTacotronV1Model(d, weights=got)
    ref = fresh.predict(feats)
    np.testing.assert_allclose(_np(pred["mel"]), _np(ref["mel"]), atol=1e-6)
    # and training goes on afterwards (optimizer state carried over the engine rebuild)
    res = model.model_fn({"source": ids, "source_length": np.array([9, 6])}, _dec_case(d, 2, 9, 3, 3)[2], ModeKeys.TRAIN, hp)
    assert res["global_step"] == 3 and np.isfinite(float(res["loss"]))
    fresh.close()
    model.close()


# ---- data parallel: SingleSpeakerTacotronV1Model.train_step(group=...) ----
def _v1dp_case():
    from tests.test_gpu_v1 import DIMS
    d = DIMS["ref"].replace(num_symbols=40, max_iters=6, min_iters=2)
    w = init_weights_v1(d, 23, perturb=True)
    B, T, S = 6, 10, 4
    ids = np.random.default_rng(8).integers(0, d.num_symbols, (B, T))
    _, lens, labels = _dec_case(d, B, T, S, 91)
    return d, w, ids, lens, labels


def _v1dp_rank(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tacotron2_amd.tacotron_v1 import SingleSpeakerTacotronV1Model
        d, w, ids, lens, labels = _v1dp_case()
        rows = slice(3 * rank, 3 * rank + 3)
        model = SingleSpeakerTacotronV1Model(d, weights=w, seed=100)
        res = model.train_step({"source": ids[rows], "source_length": lens[rows]}, {k: v[rows] for k, v in labels.items()},
                               {"initial_learning_rate": 0.002, "decay_learning_rate": False}, group=dist.group.WORLD)
        torch.cuda.synchronize()
        q.put((rank, float(res["loss"]), _np(model.flat_grads), _np(model.flat_params), model.engine.get_stat("train_persist_launches")))
        dist.barrier()
        model.close()
    finally:
        dist.destroy_process_group()


def test_single_speaker_v1_training_step_two_ranks():
    """Data parallel for the whole V1 model through the product's own path: two processes (gloo, both on this one
    GPU; their persistent launches are 160 workgroups each, so the second one's tail waits for the first inside the bounded
    spins) run SingleSpeakerTacotronV1Model.train_step(group=...) on three utterances each.  Masks come from the rank-folded
    seed, each rank differentiates its numerators / the GLOBAL non-zero weight counts, gradients are summed: the summed gradient
    equals the sum of the oracle's per-rank gradients, the loss is sum of numerators / global counts, replicas stay identical."""
    import socket
    import torch.multiprocessing as mp
    from tacotron2_amd import training as TR
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_v1dp_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, w, ids, lens, labels = _v1dp_case()
    counts = (float((labels["spec_loss_mask"] != 0).sum() * d.num_mels), float((labels["binary_loss_mask"] != 0).sum()))
    grads, loss = {}, 0.0
    for r in range(2):
        rows = slice(3 * r, 3 * r + 3)
        seed = TR.rank_seed(100, r, 2)
        lab = {k: v[rows] for k, v in labels.items()}
        enc = VT.encoder_train_forward_backward(d, w, ids[rows], None, np.zeros((3, ids.shape[1], d.cbhg_out_units), np.float32), 0.5, seed)
        dec = VT.decoder_train_forward_backward(d, w, enc["memory"], lens[rows], lab, 0.5, seed, loss_counts=counts)
        enc = VT.encoder_train_forward_backward(d, w, ids[rows], None, dec["dmemory"], 0.5, seed)
        loss += dec["loss"]                                                  # numerators / the global counts
        for k, g in list(dec["grads"].items()) + list(enc["grads"].items()):
            grads[k] = grads.get(k, 0) + g
    names = sorted(k for k in w if not k.startswith("v1/postnet/"))
    spec = TR.flatten_spec({k: w[k] for k in names})
    np.testing.assert_array_equal(res[0][3], res[1][3])                      # identical replicas after the step
    assert res[0][4] >= 2 and res[1][4] >= 2                                 # both ranks ran their recurrences as persistent launches
    for r in res:
        assert abs(r[1] - loss) <= 5e-4 * max(1.0, abs(loss)), (r[1], loss)
        for name, shape, off, size in spec:
            if name.endswith(("moving_mean", "moving_variance")):
                continue
            g_ref = np.asarray(grads[name], np.float32).reshape(-1)
            scale = float(np.abs(g_ref).max())
            assert float(np.abs(r[2][off:off + size] - g_ref).max()) <= 5e-3 * scale + 1e-9, name


# ---- block-level mirrors in training mode: DecoderV1(is_training=True, drop_rate > 0), CBHG(is_training=True) ----
def test_decoder_v1_block_in_training_mode_with_dropout():
    """DecoderV1.call(is_training=True) with drop_rate 0.5 (tacotron_v1.py:94-153, modules.py:90): the block runs the training
    forward on the flat buffer it is given; frames, stop logits and alignments against the oracle with the same masks; the
    seed advances per call; the trainer it holds accepts the backward pass."""
    import torch
    from tests.test_gpu_v1 import DIMS
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import DecoderV1, DecoderV1Trainer, EngineV1
    d = DIMS["ref"]
    B, T, S = 6, 17, 5
    w = init_weights_v1(d, 9, perturb=True)
    eng = EngineV1(d, w)
    dw = DecoderV1Trainer.decoder_weights(w)
    spec = TR.flatten_spec(dw)
    flat = torch.tensor(TR.flatten(dw, spec), device="cuda")
    mem, lens, labels = _dec_case(d, B, T, S, 5)
    kw = dict(prenet_out_units=d.decoder_prenet_out_units, attention_out_units=d.attention_out_units, decoder_out_units=d.decoder_out_units,
              num_mels=d.num_mels, outputs_per_step=d.outputs_per_step, max_iters=d.max_iters, n_feed_frame=d.n_feed_frame, engine=eng)
    with pytest.raises(ValueError):
        DecoderV1(drop_rate=0.5, **kw)(mem, is_training=True, memory_sequence_length=lens, target=labels["mel"])
    dec = DecoderV1(drop_rate=0.5, flat_params=flat, spec=spec, seed=31, **kw)
    for call in range(2):
        ref = VT.decoder_train_forward_backward(d, w, mem, lens, labels, 0.5, 31 + call)
        mel, stop, state = dec(mem, is_training=True, memory_sequence_length=lens, target=labels["mel"])
        np.testing.assert_allclose(_np(mel), ref["mel"], atol=ATOL)
        np.testing.assert_allclose(_np(stop)[..., 0], ref["stop"], atol=ATOL)
        np.testing.assert_allclose(_np(state[0].alignment_history.stack()), ref["alignments"], atol=ATOL)
    grads = torch.zeros_like(flat)
    dmem, losses = dec.trainer.backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], grads)
    assert abs(float(losses[2]) - ref["loss"]) <= 2e-4 * max(1.0, ref["loss"])
    scale = float(np.abs(ref["dmemory"]).max())
    assert float(np.abs(_np(dmem) - ref["dmemory"]).max()) <= 2e-3 * scale + 1e-9
    dec.trainer.close()
    eng.close()


@pytest.mark.parametrize("which", [0, 1])
def test_cbhg_block_in_training_mode(which):
    """CBHG.call(is_training=True) (modules.py:205-229, batch statistics) as a block, the encoder's instance with lengths and the
    PostNet's without, against the fp64 oracle."""
    import torch
    from oracle.v1_torch_train import _t, cbhg_train
    from tacotron2_amd import training as TR
    from tacotron2_amd.tacotron_v1 import CBHG, EncoderV1Trainer, EngineV1, PostNetTrainer
    d = ENC if which == 0 else POST
    w = init_weights_v1(d, 21, perturb=True)
    eng = EngineV1(d, w)
    rng = np.random.default_rng(3 + which)
    B, T = 5, 19
    if which == 0:
        tw = {k: v for k, v in w.items() if k == "embedding" or k.startswith("v1/encoder/")}
        cin, out_units, pfx, K, nh = d.encoder_prenet_out_units[1], d.cbhg_out_units, "v1/encoder/cbhg", d.max_filter_width, d.num_highway
        lens = rng.integers(T // 2, T + 1, B).astype(np.int64); lens[0] = T
    else:
        tw = PostNetTrainer.postnet_weights(w)
        cin, out_units, pfx, K, nh = d.num_mels, d.post_net_cbhg_out_units, "v1/postnet/cbhg", d.post_net_max_filter_width, d.post_net_num_highway
        lens = None
    spec = TR.flatten_spec(tw)
    flat = torch.tensor(TR.flatten(tw, spec), device="cuda")
    x = (rng.standard_normal((B, T, cin)) * 0.5).astype(np.float32)
    with pytest.raises(ValueError):
        CBHG(out_units, 0, 0, 0, 0, 0, is_training=True, engine=eng, which=which)
    blk = CBHG(out_units, 0, 0, 0, 0, 0, is_training=True, engine=eng, which=which, flat_params=flat, spec=spec)
    got = _np(blk(x, input_lengths=lens))
    W = {k: _t(v) for k, v in tw.items()}
    ref = cbhg_train(W, pfx, _t(x), K, nh, d.bn_eps, {}, False, lens).numpy()
    assert got.shape == ref.shape == (B, T, out_units)
    np.testing.assert_allclose(got, ref, atol=ATOL)
    # the inference-mode block (moving statistics) gives something else: the switch is live
    inf = _np(CBHG(out_units, 0, 0, 0, 0, 0, is_training=False, engine=eng, which=which)(x, input_lengths=lens))
    assert float(np.abs(inf - got).max()) > 1e-3
    blk.trainer.close()
    eng.close()
