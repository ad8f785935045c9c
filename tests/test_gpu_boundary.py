"""The reference's RNNCell / layer protocol at every level of the V2 cell stack, built from the public
constructors only: DecoderRNNV2.call, AttentionRNN.call and its inner wrappers, ZoneoutLSTMCell.__call__, PreNet.call,
EncoderV2 / PostNetV2(is_training=True), and every field of the final_state tuple -- each against the oracle."""
import numpy as np
import pytest
import torch

from oracle import t2_oracle as O
from oracle import torch_train as TT
from tacotron2_amd.hparams import PROFILE_P, T2Dims
from tacotron2_amd.weights import init_weights

pytestmark = pytest.mark.gpu
ATOL = 2e-4


def _np(t):
    return t.detach().cpu().numpy()


MID = T2Dims(num_symbols=30, embedding_dim=16, enc_conv_layers=2, enc_kernel=5, enc_units=16, prenet_units=(24, 12),
             attention_lstm_units=20, decoder_lstm_units=28, attention_dim=10, attention_filters=6, attention_kernel=7,
             num_mels=6, outputs_per_step=2, n_feed_frame=1, postnet_layers=2, postnet_kernel=5, postnet_channels=8,
             zoneout_cell=0.1, zoneout_output=0.1)


def _stack(d, w, memory, lens):
    """tacotron_v2.py / rnn_wrappers.py constructors only."""
    from tacotron2_amd.modules import PreNet, ZoneoutLSTMCell
    from tacotron2_amd.rnn_wrappers import AttentionRNN
    from tacotron2_amd.tacotron_v2 import DecoderRNNV2, LocationSensitiveAttention
    prenets = tuple(PreNet(u, False, 0.0) for u in d.prenet_units)
    mech = LocationSensitiveAttention(d.attention_dim, torch.from_numpy(memory).cuda(), lens, d.attention_kernel, d.attention_filters,
                                      cumulative_weights=d.cumulative_weights)
    att = AttentionRNN(ZoneoutLSTMCell(d.attention_lstm_units, False, d.zoneout_cell, d.zoneout_output), prenets, mech)
    dec = DecoderRNNV2(d.decoder_lstm_units, att, False, d.zoneout_cell, d.zoneout_output)
    att.weights = dec.weights = w
    return att, dec


def _case(seed=3, B=3, T=11):
    d = MID
    w = init_weights(d, seed, perturb=True)
    rng = np.random.default_rng(seed)
    memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
    lens = np.array([T, T - 4, T - 2][:B])
    frames = rng.uniform(-1, 1, (4, B, d.num_mels * d.n_feed_frame)).astype(np.float32)
    return d, w, memory, lens, frames


def _oracle_steps(d, w, memory, lens, frames):
    cell = O.DecoderCellV2(d, w, memory, lens)
    st = cell.zero_state()
    trace = []
    for x in frames:
        prev = st
        _, st = cell(x, st)
        trace.append((prev, st))
    return trace


def _raw_output(prev_h, new_h, zh):
    """ZoneoutLSTMCell's state h is (1 - zh) * output + zh * prev_h (modules.py:278-279): recover the raw output."""
    return (new_h - zh * prev_h) / (1.0 - zh)


def _check_state(get, st_o, T):
    att_o, s1, s2 = st_o
    for name, want in (("attention_c", att_o.cell_state[0]), ("attention_h", att_o.cell_state[1]), ("attention", att_o.attention),
                       ("alignments", att_o.alignments), ("attention_state", att_o.attention_state),
                       ("lstm1_c", s1[0]), ("lstm1_h", s1[1]), ("lstm2_c", s2[0]), ("lstm2_h", s2[1])):
        np.testing.assert_allclose(_np(get(name)), want, atol=ATOL, err_msg=name)


def test_decoder_rnn_v2_call_and_state_fields():
    d, w, memory, lens, frames = _case()
    _, dec = _stack(d, w, memory, lens)
    assert dec.output_size == d.decoder_lstm_units
    ss = dec.state_size
    assert ss[0].attention == d.enc_units and ss[0].alignments == memory.shape[1]
    state = dec.zero_state(memory.shape[0])
    with pytest.raises(RuntimeError):
        state.get("attention")                      # nothing to read before the first call
    trace = _oracle_steps(d, w, memory, lens, frames)
    for x, (prev, st_o) in zip(frames, trace):
        out, state2 = dec(torch.from_numpy(x).cuda(), state)
        assert state2 is state
        assert tuple(out.shape) == (memory.shape[0], d.decoder_lstm_units)
        np.testing.assert_allclose(_np(out), _raw_output(prev[2][1], st_o[2][1], d.zoneout_output), atol=5 * ATOL)
        np.testing.assert_allclose(_np(state.get("lstm1_output")), _raw_output(prev[1][1], st_o[1][1], d.zoneout_output), atol=5 * ATOL)
        _check_state(state.get, st_o, memory.shape[1])


def test_attention_rnn_call_and_inner_wrappers():
    d, w, memory, lens, frames = _case(seed=4)
    trace = _oracle_steps(d, w, memory, lens, frames)
    A = d.attention_lstm_units

    def want(prev, st_o):
        return _raw_output(prev[0].cell_state[1], st_o[0].cell_state[1], d.zoneout_output), st_o[0].attention

    # AttentionRNN.call: concat(attention cell output, NEW context)
    att, _ = _stack(d, w, memory, lens)
    assert att.output_size == A + d.enc_units
    state = att.zero_state(memory.shape[0])
    for x, (prev, st_o) in zip(frames, trace):
        out, _ = att(torch.from_numpy(x).cuda(), state)
        y, ctx = want(prev, st_o)
        np.testing.assert_allclose(_np(out), np.concatenate([y, ctx], axis=-1), atol=5 * ATOL)
        np.testing.assert_allclose(_np(state.get("alignments")), st_o[0].alignments, atol=ATOL)
        np.testing.assert_allclose(_np(state.get("attention_state")), st_o[0].attention_state, atol=ATOL)
    # the wrappers AttentionRNN is made of: ConcatOutputAndAttentionWrapper(DecoderPreNetWrapper(AttentionWrapper))
    att2, _ = _stack(d, w, memory, lens)
    concat_cell = att2._cell
    prenet_cell = concat_cell._cell
    assert concat_cell.output_size == A + d.enc_units and prenet_cell.output_size == A
    s_c, s_p = concat_cell.zero_state(memory.shape[0]), None
    prev, st_o = trace[0]
    out, _ = concat_cell(torch.from_numpy(frames[0]).cuda(), s_c)
    y, ctx = want(prev, st_o)
    np.testing.assert_allclose(_np(out), np.concatenate([y, ctx], axis=-1), atol=5 * ATOL)
    att3, _ = _stack(d, w, memory, lens)
    prenet_cell = att3._cell._cell
    s_p = prenet_cell.zero_state(memory.shape[0])
    out, _ = prenet_cell(torch.from_numpy(frames[0]).cuda(), s_p)
    np.testing.assert_allclose(_np(out), y, atol=5 * ATOL)
    with pytest.raises(RuntimeError):
        prenet_cell._cell.call(None, None)          # the bare AttentionWrapper runs only inside its owners


def test_unowned_inner_wrapper_is_rejected():
    from tacotron2_amd.modules import PreNet
    from tacotron2_amd.rnn_wrappers import ConcatOutputAndAttentionWrapper, DecoderPreNetWrapper
    w = DecoderPreNetWrapper(None, (PreNet(4, False, 0.0),))
    with pytest.raises(RuntimeError):
        w.zero_state(1)
    with pytest.raises(RuntimeError):
        ConcatOutputAndAttentionWrapper(w).zero_state(1)


def test_zoneout_lstm_cell_and_prenet_stand_alone():
    from tacotron2_amd.modules import PreNet, ZoneoutLSTMCell
    rng = np.random.default_rng(8)
    for B, n_in, H, zc, zh in ((1, 5, 4, 0.0, 0.0), (7, 33, 20, 0.1, 0.2), (64, 768, 1024, 0.1, 0.1)):
        kernel = rng.normal(0, 0.3 / np.sqrt(n_in + H), (n_in + H, 4 * H)).astype(np.float32)
        bias = rng.normal(0, 0.1, 4 * H).astype(np.float32)
        cell = ZoneoutLSTMCell(H, False, zc, zh, weights={"kernel": kernel, "bias": bias})
        c, h = cell.zero_state(B)
        c_o, h_o = np.zeros((B, H), np.float32), np.zeros((B, H), np.float32)
        for _ in range(3):
            x = rng.uniform(-1, 1, (B, n_in)).astype(np.float32)
            out, (c, h) = cell(torch.from_numpy(x).cuda(), (c, h))
            out_o, (c_o, h_o) = O.zoneout_lstm_cell(x, (c_o, h_o), kernel, bias, zc, zh)
            np.testing.assert_allclose(_np(out), out_o, atol=ATOL)
            np.testing.assert_allclose(_np(c), c_o, atol=ATOL)
            np.testing.assert_allclose(_np(h), h_o, atol=ATOL)
    # the sampled branch (is_training=True, modules.py:274-276): masks from the shared counter-based generator
    B, n_in, H, zc, zh = 5, 9, 12, 0.3, 0.2
    kernel = rng.normal(0, 0.3, (n_in + H, 4 * H)).astype(np.float32)
    bias = rng.normal(0, 0.1, 4 * H).astype(np.float32)
    cell = ZoneoutLSTMCell(H, True, zc, zh, weights={"kernel": kernel, "bias": bias})
    cell.seed, cell.stream = 77, 12
    c, h = cell.zero_state(B)
    c_o, h_o = np.zeros((B, H), np.float32), np.zeros((B, H), np.float32)
    for step in range(3):
        x = rng.uniform(-1, 1, (B, n_in)).astype(np.float32)
        out, (c, h) = cell(torch.from_numpy(x).cuda(), (c, h))
        out_o, (c_o, h_o) = O.zoneout_lstm_cell_train(x, (c_o, h_o), kernel, bias, zc, zh, O.rng_u01(77, 12, step, B, H), O.rng_u01(77, 13, step, B, H))
        np.testing.assert_allclose(_np(out), out_o, atol=ATOL)
        np.testing.assert_allclose(_np(c), c_o, atol=ATOL)
        np.testing.assert_allclose(_np(h), h_o, atol=ATOL)
    assert cell.time == 3
    with pytest.raises(ValueError):
        ZoneoutLSTMCell(4, False, 1.5, 0.1)
    for rows, n_in, n_out in (((3,), 6, 8), ((64,), 80, 256), ((5, 7), 9, 12)):
        k = rng.normal(0, 0.3, (n_in, n_out)).astype(np.float32)
        b = rng.normal(0, 0.1, n_out).astype(np.float32)
        x = rng.uniform(-1, 1, rows + (n_in,)).astype(np.float32)
        y = PreNet(n_out, False, 0.5, weights={"kernel": k, "bias": b})(torch.from_numpy(x).cuda())
        assert tuple(y.shape) == rows + (n_out,)
        np.testing.assert_allclose(_np(y), O.dense(x, k, b, O.relu), atol=ATOL)
    # active dropout (is_training or apply_dropout_on_inference, modules.py:90,96-98)
    k = rng.normal(0, 0.3, (7, 10)).astype(np.float32)
    b = rng.normal(0, 0.1, 10).astype(np.float32)
    x = rng.uniform(-1, 1, (6, 7)).astype(np.float32)
    for kw in ({"is_training": True}, {"is_training": False, "apply_dropout_on_inference": True}):
        pn = PreNet(10, drop_rate=0.4, weights={"kernel": k, "bias": b}, **kw)
        pn.seed, pn.stream = 5, 2
        for step in range(2):
            y = pn(torch.from_numpy(x).cuda())
            np.testing.assert_allclose(_np(y), O.dropout_train(O.dense(x, k, b, O.relu), 0.4, O.rng_u01(5, 2, step, 6, 10)), atol=ATOL)


def test_encoder_v2_training_mode_against_autograd_oracle():
    """EncoderV2(is_training=True): batch-statistics BN + dropout (modules.py:149-157) and sampled zoneout (:274-276)."""
    from tacotron2_amd.engine import Engine
    from tacotron2_amd.tacotron_v2 import EncoderV2
    d = MID
    w = init_weights(d, 6, perturb=True)
    eng = Engine(d, w)
    try:
        rng = np.random.default_rng(2)
        B, T = 4, 13
        lens = np.array([13, 7, 10, 13])
        emb = rng.normal(0, 0.5, (B, T, d.embedding_dim)).astype(np.float32)
        enc = EncoderV2(d.enc_conv_layers, d.enc_kernel, d.enc_units, 0.5, d.zoneout_cell, d.zoneout_output, True, engine=eng)
        enc.seed = 77
        W = {k: TT._t(v) for k, v in w.items()}
        for call in range(2):                                         # the seed advances per call: two different mask sets
            out = enc(torch.from_numpy(emb).cuda(), lens)
            want, stats = TT.encoder_train(d, W, TT._t(emb), lens, 77 + call, 0.5)
            np.testing.assert_allclose(_np(out), want.numpy(), atol=ATOL)
            for i in range(d.enc_conv_layers):
                m, v = stats[f"encoder/conv1d_{i}"]
                np.testing.assert_allclose(_np(enc.batch_statistics[i, 0]), m.numpy(), atol=ATOL)
                np.testing.assert_allclose(_np(enc.batch_statistics[i, 1]), v.numpy(), atol=ATOL)
        assert enc.seed == 79
        # and the same object in inference mode is the moving-statistics encoder
        enc_i = EncoderV2(d.enc_conv_layers, d.enc_kernel, d.enc_units, 0.5, d.zoneout_cell, d.zoneout_output, False, engine=eng)
        np.testing.assert_allclose(_np(enc_i(torch.from_numpy(emb).cuda(), lens)), O.encoder_v2(d, w, emb, lens), atol=ATOL)
    finally:
        eng.close()


@pytest.mark.parametrize("mode", ["free", "teacher", "generic"])
def test_decoder_v2_final_state_fields(mode):
    """final_state of DecoderV2.call is the MultiRNNCell state tuple: every field against the oracle's final state."""
    from tacotron2_amd.tacotron_v2 import DecoderV2
    d, w, memory, lens, _ = _case(seed=5)
    d = d.replace(max_iters=9, min_iters=100)
    dec = DecoderV2(prenet_out_units=d.prenet_units, drop_rate=0.0, attention_out_units=d.attention_lstm_units,
                    decoder_out_units=d.decoder_lstm_units, num_mels=d.num_mels, outputs_per_step=d.outputs_per_step, max_iters=9,
                    n_feed_frame=d.n_feed_frame, attention_dim=d.attention_dim, attention_kernel=d.attention_kernel,
                    attention_filters=d.attention_filters, zoneout_factor_cell=d.zoneout_cell, zoneout_factor_output=d.zoneout_output,
                    min_iters=100, weights=w)
    mem_t = torch.from_numpy(memory).cuda()
    if mode == "teacher":
        tgt = np.random.default_rng(1).uniform(0, 1, (memory.shape[0], 10, d.num_mels)).astype(np.float32)
        _, _, st_o = O.decoder_v2(d, w, memory, is_training=True, memory_sequence_length=lens, target=tgt)
        _, _, st = dec(mem_t, is_training=False, is_validation=True, teacher_forcing=True, memory_sequence_length=lens, target=tgt)
    else:
        _, _, st_o = O.decoder_v2(d, w, memory, memory_sequence_length=lens)
        _, _, st = dec(mem_t, is_training=False, memory_sequence_length=lens, generic_loop=(mode == "generic"))
    fields = {"attention_c": st[0].cell_state.c, "attention_h": st[0].cell_state.h, "attention": st[0].attention,
              "alignments": st[0].alignments, "attention_state": st[0].attention_state,
              "lstm1_c": st[1].c, "lstm1_h": st[1].h, "lstm2_c": st[2].c, "lstm2_h": st[2].h}
    _check_state(lambda n: fields[n], st_o, memory.shape[1])
    assert st[0].time == st_o[0].time == len(st_o[0].alignment_history)
    np.testing.assert_allclose(_np(st[0].alignment_history.stack()), np.stack(st_o[0].alignment_history), atol=ATOL)


@pytest.mark.parametrize("persist", [1, 0])
def test_final_state_after_persistent_decode_at_paper_sizes(persist):
    """The one-launch decoder leaves the same state tuple as the launch-per-phase schedule and the oracle."""
    from tacotron2_amd.engine import Engine
    from tacotron2_amd.tacotron_v2 import DecoderV2
    d = PROFILE_P
    w = init_weights(d, seed=91, perturb=True)
    eng = Engine(d, w)
    try:
        eng.set_option("persist", persist)
        rng = np.random.default_rng(12)
        B, T, S = 5, 37, 6
        lens = np.array([37, 20, 33, 37, 9])
        memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
        memory *= (np.arange(T)[None, :] < lens[:, None])[:, :, None]
        dec = DecoderV2(prenet_out_units=d.prenet_units, drop_rate=0.0, attention_out_units=d.attention_lstm_units,
                        decoder_out_units=d.decoder_lstm_units, num_mels=d.num_mels, outputs_per_step=d.outputs_per_step, max_iters=S,
                        n_feed_frame=d.n_feed_frame, attention_dim=d.attention_dim, attention_kernel=d.attention_kernel,
                        attention_filters=d.attention_filters, zoneout_factor_cell=d.zoneout_cell,
                        zoneout_factor_output=d.zoneout_output, min_iters=100, engine=eng)
        before = eng.get_stat("persist_launches")
        _, stop, st = dec(torch.from_numpy(memory).cuda(), is_training=False, memory_sequence_length=lens)
        assert eng.get_stat("persist_launches") == before + persist
        assert stop.shape[1] == S
        _, _, st_o = O.decoder_v2(d.replace(min_iters=100), w, memory, memory_sequence_length=lens, max_iters=S)
        fields = {"attention_c": st[0].cell_state.c, "attention_h": st[0].cell_state.h, "attention": st[0].attention,
                  "alignments": st[0].alignments, "attention_state": st[0].attention_state,
                  "lstm1_c": st[1].c, "lstm1_h": st[1].h, "lstm2_c": st[2].c, "lstm2_h": st[2].h}
        _check_state(lambda n: fields[n], st_o, T)
    finally:
        eng.close()


@pytest.mark.parametrize("dims,B,T,S", [("mid", 3, 11, 5), ("paper", 5, 37, 6)])
def test_decoder_v2_training_mode_samples_dropout_and_zoneout(dims, B, T, S):
    """DecoderV2.call(is_training=True) with drop_rate / zoneout factors > 0 (tacotron_v1.py:134-150, modules.py:90,96-98,
    274-276): the library's training forward (one persistent launch at paper sizes), masks from the shared counter-based
    generator; outputs, alignment history and the final state tuple against the torch oracle; the seed advances per call."""
    from tacotron2_amd.engine import Engine
    from tacotron2_amd.tacotron_v2 import DecoderV2
    d = MID if dims == "mid" else PROFILE_P
    w = init_weights(d, seed=17, perturb=True)
    eng = Engine(d, w)
    try:
        rng = np.random.default_rng(4)
        lens = np.array([T, T - 4, T - 2, T, 9][:B])
        memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
        r, M = d.outputs_per_step, d.num_mels
        tgt = rng.uniform(-1, 1, (B, S * r, M)).astype(np.float32)
        dec = DecoderV2(prenet_out_units=d.prenet_units, drop_rate=0.5, attention_out_units=d.attention_lstm_units,
                        decoder_out_units=d.decoder_lstm_units, num_mels=M, outputs_per_step=r, max_iters=200,
                        n_feed_frame=d.n_feed_frame, attention_dim=d.attention_dim, attention_kernel=d.attention_kernel,
                        attention_filters=d.attention_filters, zoneout_factor_cell=d.zoneout_cell,
                        zoneout_factor_output=d.zoneout_output, engine=eng, seed=300)
        W = {k: TT._t(v) for k, v in w.items()}
        before = eng.get_stat("persist_launches")
        for call in range(2):
            mel, stop, st = dec(torch.from_numpy(memory).cuda(), is_training=True, memory_sequence_length=lens, target=tgt)
            mel_o, stop_o, align_o, st_o = TT.decoder_train(d, W, TT._t(memory), lens, TT._t(tgt), 300 + call, 0.5, return_state=True)
            np.testing.assert_allclose(_np(mel), mel_o.numpy(), atol=ATOL)
            np.testing.assert_allclose(_np(stop), stop_o.numpy(), atol=ATOL)
            np.testing.assert_allclose(_np(st[0].alignment_history.stack()), align_o.numpy(), atol=ATOL)
            fields = {"attention_c": st[0].cell_state.c, "attention_h": st[0].cell_state.h, "attention": st[0].attention,
                      "alignments": st[0].alignments, "attention_state": st[0].attention_state,
                      "lstm1_c": st[1].c, "lstm1_h": st[1].h, "lstm2_c": st[2].c, "lstm2_h": st[2].h}
            for name, got in fields.items():
                np.testing.assert_allclose(_np(got), st_o[name].numpy(), atol=ATOL, err_msg=name)
            assert st[0].time == S
        assert eng.get_stat("persist_launches") == before + (2 if dims == "paper" else 0)
        # the masks matter: the inference arithmetic on the same inputs differs
        mel_inf, _, _ = dec(torch.from_numpy(memory).cuda(), is_training=False, is_validation=True, teacher_forcing=True,
                            memory_sequence_length=lens, target=tgt)
        assert np.abs(_np(mel_inf) - _np(mel)).max() > 1e-3
        # maximum_iterations=max_iters truncates the loop (tacotron_v1.py:150)
        dec.max_iters = S - 2
        mel_s, stop_s, st_s = dec(torch.from_numpy(memory).cuda(), is_training=True, memory_sequence_length=lens, target=tgt)
        assert mel_s.shape == (B, (S - 2) * r, M) and stop_s.shape == (B, S - 2, 1) and st_s[0].time == S - 2
    finally:
        eng.close()


def test_gru_cell_factory_describes_the_v1_cell():
    from tacotron2_amd.rnn_impl import GRUImpl, gru_cell_factory
    cell = gru_cell_factory(GRUImpl.GRUCell, 12)
    assert cell.num_units == 12 and cell.state_size == 12 and cell.output_size == 12


def test_postnet_v2_training_mode_against_autograd_oracle():
    """PostNetV2(is_training=True): batch-statistics BN, tanh (none on the last convolution), dropout (modules.py:149-157),
    Dense + residual (tacotron_v2.py:200-204)."""
    from tacotron2_amd.engine import Engine
    from tacotron2_amd.tacotron_v2 import PostNetV2
    d = MID
    w = init_weights(d, 8, perturb=True)
    eng = Engine(d, w)
    try:
        rng = np.random.default_rng(5)
        B, T = 3, 17
        mel = rng.normal(0, 0.7, (B, T, d.num_mels)).astype(np.float32)
        post = PostNetV2(d.num_mels, d.postnet_layers, d.postnet_kernel, d.postnet_channels, True, drop_rate=0.5, engine=eng)
        post.seed = 11
        W = {k: TT._t(v) for k, v in w.items()}
        for call in range(2):
            out = post(torch.from_numpy(mel).cuda())
            want, stats = TT.postnet_train(d, W, TT._t(mel), 11 + call, 0.5)
            np.testing.assert_allclose(_np(out), want.numpy(), atol=ATOL)
            for i in range(d.postnet_layers):
                m, v = stats[f"postnet/conv1d_{i + 1}"]
                np.testing.assert_allclose(_np(post.batch_statistics[i, 0]), m.numpy(), atol=ATOL)
                np.testing.assert_allclose(_np(post.batch_statistics[i, 1]), v.numpy(), atol=ATOL)
        assert post.seed == 13
        # zero drop rate: only the statistics differ from inference
        post0 = PostNetV2(d.num_mels, d.postnet_layers, d.postnet_kernel, d.postnet_channels, True, drop_rate=0.0, engine=eng)
        want0, _ = TT.postnet_train(d, W, TT._t(mel), 0, 0.0)
        np.testing.assert_allclose(_np(post0(torch.from_numpy(mel).cuda())), want0.numpy(), atol=ATOL)
        post_i = PostNetV2(d.num_mels, d.postnet_layers, d.postnet_kernel, d.postnet_channels, False, engine=eng)
        np.testing.assert_allclose(_np(post_i(torch.from_numpy(mel).cuda())), O.postnet_v2(d, w, mel), atol=ATOL)
    finally:
        eng.close()


def test_output_projection_wrapper_call_on_a_zoneout_cell():
    """rnn_wrappers.py:174-186 on its own: output, state = cell(inputs, state); tensordot(output, kernel) + bias, activation
    (missing #4): the wrapped cell runs on the library (t2_zoneout_lstm_cell), the projection on t2_dense."""
    from tacotron2_amd.modules import ZoneoutLSTMCell
    from tacotron2_amd.rnn_wrappers import OutputProjectionWrapper
    rng = np.random.default_rng(21)
    B, n_in, H, N = 6, 10, 12, 7
    kernel = rng.normal(0, 0.3, (n_in + H, 4 * H)).astype(np.float32)
    bias = rng.normal(0, 0.1, 4 * H).astype(np.float32)
    pk = rng.normal(0, 0.3, (H, N)).astype(np.float32)
    pb = rng.normal(0, 0.1, N).astype(np.float32)
    for activation, use_bias, f in ((None, True, lambda v: v), ("tanh", True, np.tanh), ("relu", False, lambda v: np.maximum(v, 0)),
                                    (torch.sigmoid, True, lambda v: 1 / (1 + np.exp(-v)))):
        cell = OutputProjectionWrapper(ZoneoutLSTMCell(H, False, 0.1, 0.2, weights={"kernel": kernel, "bias": bias}), N,
                                       activation=activation, use_bias=use_bias, weights={"kernel": pk, "bias": pb})
        assert cell.output_size == N
        state = cell.zero_state(B)
        c_o, h_o = np.zeros((B, H), np.float32), np.zeros((B, H), np.float32)
        for _ in range(3):
            x = rng.uniform(-1, 1, (B, n_in)).astype(np.float32)
            y, state = cell(torch.from_numpy(x).cuda(), state)
            out_o, (c_o, h_o) = O.zoneout_lstm_cell(x, (c_o, h_o), kernel, bias, 0.1, 0.2)
            want = f(out_o @ pk + (pb if use_bias else 0.0))
            np.testing.assert_allclose(_np(y), want, atol=ATOL)
            np.testing.assert_allclose(_np(state[1]), h_o, atol=ATOL)
    with pytest.raises(ValueError):
        OutputProjectionWrapper(ZoneoutLSTMCell(H, False, weights={"kernel": kernel, "bias": bias}), N, weights={"kernel": pk[:, :3], "bias": pb})(
            torch.zeros(B, n_in).cuda(), cell.zero_state(B))
