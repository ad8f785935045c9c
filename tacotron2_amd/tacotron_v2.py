"""Mirror of the reference's tacotron/tacotron_v2.py blocks (EncoderV2 :33, LocationSensitiveAttention
:81, DecoderRNNV2 :147, PostNetV2 :181) plus the V2 assembly the reference never wrote
(README.md:93 "Add Tacotron2 model" is an open ToDo): ``DecoderV2`` follows ``DecoderV1.call``
(tacotron_v1.py:118-153) and ``TacotronV2Model.model_fn`` follows the V1 model_fn (models.py:18-129)
with the V2 blocks swapped in.

No arithmetic lives here: everything runs in the HIP library through ``engine.Engine``.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import training
from .engine import Engine, shared_context
from .hparams import T2Dims
from .helpers import StopTokenBasedInferenceHelper, TrainingHelper, ValidationHelper
from .modules import Conv1d, PreNet, ZoneoutLSTMCell
from .rnn_impl import LSTMImpl
from .rnn_wrappers import AttentionRNN, OutputAndStopTokenWrapper
from .weights import init_weights


class ModeKeys:
    """tf.estimator.ModeKeys."""
    TRAIN, EVAL, PREDICT = "train", "eval", "infer"


class EncoderV2:
    """tacotron_v2.py:33-63.  ``__call__(inputs, input_lengths)``: inputs fp32 [B, T_in, embedding_dim]."""

    def __init__(self, num_conv_layers, kernel_size, out_units, drop_rate, zoneout_factor_cell, zoneout_factor_output,
                 is_training, lstm_impl=LSTMImpl.LSTMCell, engine: Optional[Engine] = None, name=None, dtype=None):
        assert out_units % 2 == 0
        self.out_units = out_units
        self.zoneout_factor_cell = zoneout_factor_cell
        self.zoneout_factor_output = zoneout_factor_output
        self.is_training = is_training
        self._lstm_impl = lstm_impl
        self.convolutions = [Conv1d(kernel_size, out_units, activation="relu", is_training=is_training, drop_rate=drop_rate,
                                    name=f"conv1d_{i}") for i in range(0, num_conv_layers)]
        self._cells = [ZoneoutLSTMCell(out_units // 2, is_training, zoneout_factor_cell, zoneout_factor_output,
                                       lstm_impl=lstm_impl) for _ in range(2)]
        self.drop_rate = drop_rate
        self._engine = engine
        self._trainers = {}
        self.seed = 0                   # seed of the next training-mode call's dropout / zoneout masks; advances per call
        self.batch_statistics = None    # [layers, 2, out_units] batch mean | variance of the last training-mode call

    def __call__(self, inputs, input_lengths=None):
        return self.call(inputs, input_lengths)

    def call(self, inputs, input_lengths=None):
        """Inference arithmetic (moving statistics, no dropout, deterministic zoneout), or with ``is_training`` the training
        arithmetic of Conv1d.call (modules.py:149-157: batch statistics over all positions, dropout) and the sampled zoneout
        branch of ZoneoutLSTMCell (modules.py:274-276) -- t2_train_encoder_forward_embedded."""
        if self._engine is None:
            raise RuntimeError("EncoderV2 needs an engine (see TacotronV2Model)")
        if not self.is_training:
            return self._engine.encode_embedded(inputs, input_lengths)
        x = self._engine._f32(inputs)
        key = (int(x.shape[0]), int(x.shape[1]))
        if key not in self._trainers:
            if len(self._trainers) >= 2:
                self._trainers.pop(next(iter(self._trainers))).close()
            self._trainers[key] = training.Trainer(self._engine, key[0], key[1], 1)
        memory, stats = self._trainers[key].encoder_forward_embedded(x, input_lengths, drop_rate=self.drop_rate, seed=self.seed)
        self.seed += 1
        self.batch_statistics = stats
        return memory


class LocationSensitiveAttention:
    """tacotron_v2.py:81-144 over BahdanauAttention: holds the memory and the sizes; values / keys
    (the constructor side effects of the reference) are computed by t2_state_create."""

    def __init__(self, num_units, memory, memory_sequence_length, attention_kernel, attention_filters, smoothing=False,
                 cumulative_weights=True, dtype=None, name="LocationSensitiveAttention"):
        if smoothing:
            # reference :93,143-144: _smoothing_normalization is declared without ``self`` and raises
            # TypeError the moment it is called; the build rejects it up front with the same type.
            raise TypeError("_smoothing_normalization() takes 1 positional argument but 3 were given")
        self.num_units = num_units
        self.memory = memory
        self.memory_sequence_length = memory_sequence_length
        self.attention_kernel = attention_kernel
        self.attention_filters = attention_filters
        self.cumulative_weights = cumulative_weights

    @property
    def batch_size(self):
        return self.memory.shape[0]

    @property
    def alignments_size(self):
        return self.memory.shape[1]

    @property
    def values_depth(self):
        return self.memory.shape[2]

    @property
    def state_size(self):
        return self.alignments_size


class DecoderRNNV2:
    """tacotron_v2.py:147-178: MultiRNNCell([attention_cell, ZoneoutLSTM, ZoneoutLSTM]); no residuals."""

    def __init__(self, out_units, attention_cell: AttentionRNN, is_training, zoneout_factor_cell=0.0,
                 zoneout_factor_output=0.0, lstm_impl=LSTMImpl.LSTMCell, name=None, dtype=None):
        self.attention_cell = attention_cell
        self.lstms = [ZoneoutLSTMCell(out_units, is_training, zoneout_factor_cell, zoneout_factor_output, lstm_impl=lstm_impl),
                      ZoneoutLSTMCell(out_units, is_training, zoneout_factor_cell, zoneout_factor_output, lstm_impl=lstm_impl)]
        self._out_units = out_units

    @property
    def state_size(self):
        return (self.attention_cell.state_size, self.lstms[0].state_size, self.lstms[1].state_size)

    @property
    def output_size(self):
        return self._out_units

    # -- RNNCell protocol on the cell itself (tacotron_v2.py:177-178) -----------------------------------------------
    def _bound(self, n_feed_units):
        """The HIP library runs the cell stack as one unit (t2_decode_step); a bare DecoderRNNV2 is bound through an
        OutputAndStopTokenWrapper whose projections nobody reads."""
        if getattr(self, "_outer", None) is None or self._outer._n_feed_units != n_feed_units:
            from .rnn_wrappers import OutputAndStopTokenWrapper
            unread = ("decoder/out_projection/", "decoder/stop_token_projection/")
            own = {k: v for k, v in (getattr(self, "weights", None) or {}).items() if not k.startswith(unread)}
            self._outer = OutputAndStopTokenWrapper(self, n_feed_units, weights=own, n_feed_units=n_feed_units, num_mels=n_feed_units)
        return self._outer

    def zero_state(self, batch_size, dtype=None):
        from .rnn_wrappers import DeferredState
        return DeferredState(batch_size)            # the library state is created at the first call, when the input width is known

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self.output_size)

    def call(self, inputs, state):
        """MultiRNNCell([attention_cell, ZoneoutLSTM, ZoneoutLSTM]) step: returns (raw output of the last LSTM [B, out_units],
        state); ``state`` (a DecoderState from zero_state) is advanced in place."""
        from .rnn_wrappers import DeferredState
        outer = self._bound(int(inputs.shape[-1]))
        if isinstance(state, DeferredState):
            state.bind(lambda: outer.zero_state(state.batch_size))
        outer.call(inputs, state)
        return state.get("decoder_output"), state

    __call__ = call


class PostNetV2:
    """tacotron_v2.py:181-204.  Inference arithmetic (moving statistics folded into the convolutions, no dropout), or with
    ``is_training`` the training arithmetic of Conv1d.call (modules.py:149-157: batch statistics over all positions, dropout
    sampled from the library's counter-based generator; ``seed`` advances per call, ``batch_statistics`` keeps the last
    call's mean | variance per layer).  Forward only: the reference's loss does not contain the PostNet (models.py:58-61)."""

    def __init__(self, out_units, num_postnet_layers, kernel_size, out_channels, is_training, drop_rate=0.5,
                 engine: Optional[Engine] = None, name=None, dtype=None):
        self.out_units, self.num_postnet_layers, self.kernel_size, self.out_channels = out_units, num_postnet_layers, kernel_size, out_channels
        self.is_training, self.drop_rate = is_training, drop_rate
        self._engine = engine
        self.seed = 0
        self.batch_statistics = None

    def __call__(self, inputs, **kwargs):
        return self.call(inputs)

    def call(self, inputs, **kwargs):
        if self._engine is None:
            raise RuntimeError("PostNetV2 needs an engine (see TacotronV2Model)")
        if not self.is_training:
            return self._engine.postnet(inputs)
        out, self.batch_statistics = self._engine.postnet_training(inputs, drop_rate=self.drop_rate, seed=self.seed)
        self.seed += 1
        return out


def dynamic_decode(cell, helper, initial_state, maximum_iterations=None):
    """tf.contrib.seq2seq.dynamic_decode(BasicDecoder(cell, helper, initial_state), maximum_iterations)
    with impute_finished=False, as called at tacotron_v1.py:149-150 -- the generic (one C-ABI step per
    iteration) loop.  Returns ((mel [B,S,M*r], stop [B,S,1]), final_state, steps)."""
    finished, inputs = helper.initialize()
    if maximum_iterations is not None:
        finished = finished | bool(0 >= maximum_iterations)
    state, time = initial_state, 0
    mels, stops = [], []
    while not bool(torch.all(finished)):
        outputs, state = cell(inputs, state)
        step_finished, inputs, state = helper.next_inputs(time, outputs, state, None)
        next_finished = step_finished | finished              # broadcasts the batch-coupled [1] case
        if maximum_iterations is not None:
            next_finished = next_finished | bool(time + 1 >= maximum_iterations)
        mels.append(outputs[0].clone())
        stops.append(outputs[1].clone())
        finished = next_finished
        time += 1
    return (torch.stack(mels, dim=1), torch.stack(stops, dim=1)), state, time


class _AlignmentHistory:
    def __init__(self, stacked):
        self._stacked = stacked

    def stack(self):
        return self._stacked          # [S, B, T_in], like TensorArray.stack() of the reference


class DecoderV2:
    """The V2 counterpart of DecoderV1 (tacotron_v1.py:94-153): same constructor arguments plus the
    V2 block sizes, same ``call`` signature, same helper selection (:134-147), same return triple
    ``(mel [B, S*r, M], stop [B, S, 1], final_state)`` with
    ``final_state[0].alignment_history.stack()`` -> [S, B, T_in] (models.py:53)."""

    def __init__(self, prenet_out_units=(256, 256), drop_rate=0.5, attention_out_units=1024, decoder_out_units=1024,
                 num_mels=80, outputs_per_step=2, max_iters=200, n_feed_frame=1, attention_dim=128, attention_kernel=31,
                 attention_filters=32, cumulative_weights=True, zoneout_factor_cell=0.0, zoneout_factor_output=0.0,
                 min_iters=10, engine: Optional[Engine] = None, weights: Optional[Dict[str, np.ndarray]] = None, seed=1234,
                 name=None, dtype=None):
        self._prenet_out_units = tuple(prenet_out_units)
        self._drop_rate = drop_rate
        self.attention_out_units = attention_out_units
        self.decoder_out_units = decoder_out_units
        self.num_mels = num_mels
        self.outputs_per_step = outputs_per_step
        self.max_iters = max_iters
        self.min_iters = min_iters
        self.n_feed_frame = n_feed_frame
        self.attention_dim, self.attention_kernel, self.attention_filters = attention_dim, attention_kernel, attention_filters
        self.cumulative_weights = cumulative_weights
        self.zoneout_factor_cell, self.zoneout_factor_output = zoneout_factor_cell, zoneout_factor_output
        self._engine, self._weights, self._seed = engine, weights, seed
        self.seed = seed & 0xFFFFFFFF       # mask seed of the next training-mode call (advances by one per call)

    def _cell_for(self, source, is_training, memory_sequence_length, allow_sampling=False):
        prenets = tuple(PreNet(u, is_training, self._drop_rate) for u in self._prenet_out_units)
        mech = LocationSensitiveAttention(self.attention_dim, source, memory_sequence_length, self.attention_kernel,
                                          self.attention_filters, cumulative_weights=self.cumulative_weights)
        att_lstm = ZoneoutLSTMCell(self.attention_out_units, is_training, self.zoneout_factor_cell, self.zoneout_factor_output)
        attention_cell = AttentionRNN(att_lstm, prenets, mech)
        decoder_cell = DecoderRNNV2(self.decoder_out_units, attention_cell, is_training, self.zoneout_factor_cell,
                                    self.zoneout_factor_output)
        cell = OutputAndStopTokenWrapper(decoder_cell, self.num_mels * self.outputs_per_step, weights=self._weights,
                                         n_feed_units=self.num_mels * self.n_feed_frame, engine=self._engine, seed=self._seed,
                                         num_mels=self.num_mels)
        cell.decoder_dims(allow_sampling=allow_sampling)   # runs the training-mode / size checks of the stack
        self._engine = cell.engine()
        return cell

    def _sampling(self, is_training):
        return bool(is_training) and ((self._drop_rate or 0) > 0 or self.zoneout_factor_cell > 0 or self.zoneout_factor_output > 0)

    def _trainer_for(self, B, T_in, S):
        key = (id(self._engine), B, T_in, S)
        if getattr(self, "_trainer_key", None) != key:
            self._trainer, self._trainer_key = training.Trainer(self._engine, B, T_in, S), key
        return self._trainer

    def __call__(self, source, **kw):
        return self.call(source, **kw)

    def call(self, source, is_training=None, is_validation=None, teacher_forcing=False, memory_sequence_length=None,
             target=None, generic_loop=False):
        assert is_training is not None
        assert self.n_feed_frame <= self.outputs_per_step            # helpers.py:55,100,144
        sampled = self._sampling(is_training) and not generic_loop
        cell = self._cell_for(source, is_training, memory_sequence_length, allow_sampling=sampled)
        eng = self._engine
        B = source.shape[0]
        M, r, nf = self.num_mels, self.outputs_per_step, self.n_feed_frame
        if sampled:
            # TrainingHelper with sampled PreNet dropout (modules.py:90,96-98) and zoneout (modules.py:274-276): the training
            # forward of the library (t2_train_decoder_forward), masks from (seed, stream, step, row, column)
            S = min(target.shape[1] // r, self.max_iters)            # maximum_iterations=max_iters (:150)
            tr = self._trainer_for(B, source.shape[1], S)
            mel_output, stop, align = tr.decoder_forward(source, memory_sequence_length, eng._f32(target)[:, :S * r].contiguous(),
                                                         prenet_drop_rate=self._drop_rate, seed=self.seed)
            self.seed = (self.seed + 1) & 0xFFFFFFFF
        elif generic_loop:
            # literal transcription of tacotron_v1.py:132-152 on the RNNCell / Helper protocol
            tgt = eng._f32(target) if target is not None else None
            helper = TrainingHelper(tgt, M, r, n_feed_frame=nf) if is_training \
                else ValidationHelper(tgt, B, M, r, n_feed_frame=nf, teacher_forcing=teacher_forcing) if is_validation \
                else StopTokenBasedInferenceHelper(B, M, r, n_feed_frame=nf, min_iters=self.min_iters, device=eng.device)
            state0 = cell.zero_state(B)
            hist = []
            base_call = cell.call
            cell.mask_seed = self.seed                              # single cell calls sample with this call's seed (t2_decode_step_train)
            if self._sampling(is_training):
                self.seed = (self.seed + 1) & 0xFFFFFFFF

            def stepping(inputs, state):
                out = base_call(inputs, state)
                hist.append(state.get("alignments"))
                return out
            (dec, stop), final, _ = dynamic_decode(stepping, helper, state0, maximum_iterations=self.max_iters)
            mel_output = dec.reshape(B, -1, M)
            align = torch.stack(hist, dim=0)
        elif is_training or is_validation:
            # TrainingHelper / ValidationHelper: S = T_out // r steps for every row
            if not is_training and not teacher_forcing:
                mel_output, stop, align = eng.decode_teacher(source, memory_sequence_length, target, teacher_forcing=False)
            else:
                mel_output, stop, align = eng.decode_teacher(source, memory_sequence_length, target, teacher_forcing=True)
            S = min(stop.shape[1], self.max_iters)                   # maximum_iterations=max_iters (:150)
            mel_output, stop, align = mel_output[:, :S * r], stop[:, :S], align[:S]
        else:
            mel_output, stop, align, _ = eng.decode_free(source, memory_sequence_length, max_iters=self.max_iters,
                                                         min_iters=self.min_iters)
        final_state = self._final_state(eng, B, source.shape[1], align, None if not generic_loop else final)
        return mel_output, stop, final_state

    @staticmethod
    def _final_state(eng, B, T_in, align, state=None):
        """The state tuple DecoderRNNV2 leaves behind (tacotron_v2.py:155-161 MultiRNNCell, state_is_tuple=True):
        (AttentionWrapperState(cell_state=LSTMStateTuple(c, h), attention, time, alignments, alignment_history, attention_state),
        LSTMStateTuple(c, h) of LSTM 1, LSTMStateTuple(c, h) of LSTM 2) -- models.py:53 reads [0].alignment_history.stack()."""
        get = (lambda n: state.get(n)) if state is not None else (lambda n: eng.decode_final_state(n, B, T_in))
        lstm = lambda p: SimpleNamespace(c=get(p + "_c"), h=get(p + "_h"))
        att = SimpleNamespace(cell_state=lstm("attention"), attention=get("attention"), time=int(align.shape[0]), alignments=get("alignments"),
                              alignment_history=_AlignmentHistory(align), attention_state=get("attention_state"))
        return (att, lstm("lstm1"), lstm("lstm2"))


def spec_loss(y_hat, y, mask, n_priority_freq=None, priority_w=0, engine=None):
    """SingleSpeakerTacotronV1Model.spec_loss (models.py:136-144) on the HIP library (t2_spec_loss)."""
    return training.spec_loss(engine or shared_context(_device_index(y_hat)), y_hat, y, mask, n_priority_freq, priority_w)


def binary_loss(done_hat, done, mask, engine=None):
    """binary_loss (models.py:146-148) on the HIP library (t2_binary_loss)."""
    return training.binary_loss(engine or shared_context(_device_index(done_hat)), done_hat, done, mask)


def _device_index(x):
    return x.device.index if isinstance(x, torch.Tensor) and x.is_cuda else torch.cuda.current_device()


learning_rate_decay = training.learning_rate_decay      # models.py:150-154


class TacotronV2Model:
    """The V2 counterpart of SingleSpeakerTacotronV1Model (models.py:15-133): one HIP model for
    embedding + EncoderV2 + decoder + PostNetV2, and ``model_fn(features, labels, mode, params)``.

    PREDICT returns the reference's prediction dict (models.py:122-129: id, mel, alignment
    [B, T_in, S], source, text) plus ``mel_postnet`` and ``stop_token``; EVAL returns the losses of
    the free-feeding and the teacher-forced pass (models.py:94-107); TRAIN runs one optimizer step
    (models.py:63-76: training-mode forward, loss, gradients, clip_by_global_norm(1.0), Adam with the
    optional Noam decay, batch-norm UPDATE_OPS) on the device and returns ``{"loss", "mel_loss",
    "done_loss", "global_step", "learning_rate"}`` -- the EstimatorSpec's loss plus what add_training_stats
    records (models.py:72).  With torch.distributed initialised the gradients are averaged over the
    data-parallel group (RCCL) before the clip; every rank then applies the same update.

    Training hyper-parameters come from ``params`` (dict or object) with the reference's defaults
    (hparams.py:44,34,64-68): decoder_prenet_drop_rate 0.5, encoder drop rate 0.5, adam_beta1 0.9,
    adam_beta2 0.999, adam_eps 1e-8, initial_learning_rate 0.002, decay_learning_rate True.  The dropout /
    zoneout masks of step n are drawn from the counter-based generator with seed ``seed + n``."""

    TRAIN_DEFAULTS = dict(decoder_prenet_drop_rate=0.5, encoder_drop_rate=0.5, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                          initial_learning_rate=0.002, decay_learning_rate=True)

    def __init__(self, params: T2Dims, weights: Optional[Dict[str, np.ndarray]] = None, device: int = 0, seed: int = 1234):
        self.params = params
        self.weights = weights if weights is not None else init_weights(params, seed)
        self.engine = Engine(params, self.weights, device=device)
        p = params
        self.decoder = DecoderV2(prenet_out_units=p.prenet_units, drop_rate=0.0, attention_out_units=p.attention_lstm_units,
                                 decoder_out_units=p.decoder_lstm_units, num_mels=p.num_mels,
                                 outputs_per_step=p.outputs_per_step, max_iters=p.max_iters, n_feed_frame=p.n_feed_frame,
                                 attention_dim=p.attention_dim, attention_kernel=p.attention_kernel,
                                 attention_filters=p.attention_filters, cumulative_weights=p.cumulative_weights,
                                 zoneout_factor_cell=p.zoneout_cell, zoneout_factor_output=p.zoneout_output,
                                 min_iters=p.min_iters, engine=self.engine)
        self.postnet = PostNetV2(p.num_mels, p.postnet_layers, p.postnet_kernel, p.postnet_channels, is_training=False,
                                 engine=self.engine)
        self.global_step = 0            # tf.train.get_global_step()
        self.seed = seed
        self._train = None              # flat parameters / gradients / optimizer, built by the first TRAIN call
        self._trainers = {}             # (B, T_in, S) -> training.Trainer

    def _train_state(self, hp):
        if self._train is None:
            spec = training.flatten_spec(self.weights)
            flat = torch.tensor(training.flatten(self.weights, spec), device=self.engine.device)
            lr0 = float(hp["initial_learning_rate"])
            lr = (lambda step: training.learning_rate_decay(lr0, step)) if hp["decay_learning_rate"] else lr0
            opt = training.AdamOptimizer(self.engine, flat.numel(), learning_rate=lr, beta1=hp["adam_beta1"], beta2=hp["adam_beta2"],
                                         epsilon=hp["adam_eps"])
            self._train = SimpleNamespace(spec=spec, flat=flat, grads=torch.zeros_like(flat), opt=opt)
        return self._train

    def _trainer(self, B, T_in, S):
        key = (int(B), int(T_in), int(S))
        if key not in self._trainers:
            if len(self._trainers) >= 4:                 # tapes are large: keep the most recent shapes only
                self._trainers.pop(next(iter(self._trainers))).close()
            tr = training.Trainer(self.engine, *key)
            tr.bind_gradients(self._train.spec)
            self._trainers[key] = tr
        return self._trainers[key]

    def parameters(self) -> Dict[str, np.ndarray]:
        """The current variables (TF names and layouts), read back from the flat device buffer after training."""
        if self._train is None:
            return self.weights
        return training.unflatten(self._train.flat.cpu().numpy(), self._train.spec)

    def _train_step(self, f, lab, params):
        hp = dict(self.TRAIN_DEFAULTS)
        if params is not None:
            src = params if isinstance(params, dict) else params.__dict__
            hp.update({k: src[k] for k in hp if k in src})
        st = self._train_state(hp)
        source = np.asarray(f["source"].cpu() if isinstance(f["source"], torch.Tensor) else f["source"])
        lens = np.asarray(f["source_length"].cpu() if isinstance(f["source_length"], torch.Tensor) else f["source_length"])
        mel = self.engine._f32(lab["mel"])
        r = self.params.outputs_per_step
        if mel.shape[1] % r != 0:
            raise ValueError("target length must be a multiple of outputs_per_step (the reference's padded batches are)")
        tr = self._trainer(source.shape[0], source.shape[1], mel.shape[1] // r)
        labels = {"mel": mel, "done": lab["done"], "spec_loss_mask": lab["spec_loss_mask"], "binary_loss_mask": lab["binary_loss_mask"]}
        step0 = self.global_step
        lr = st.opt.learning_rate(step0) if callable(st.opt.learning_rate) else st.opt.learning_rate
        self.global_step, losses = tr.train_step(st.opt, st.flat, st.grads, {"source": source, "source_length": lens}, labels,
                                                 global_step=step0, seed=self.seed + step0,
                                                 prenet_drop_rate=hp["decoder_prenet_drop_rate"], encoder_drop_rate=hp["encoder_drop_rate"])
        return {"loss": losses[2], "mel_loss": losses[0], "done_loss": losses[1], "global_step": self.global_step, "learning_rate": lr}

    def model_fn(self, features, labels=None, mode=ModeKeys.PREDICT, params=None):
        is_training = mode == ModeKeys.TRAIN
        is_validation = mode == ModeKeys.EVAL
        f = features if isinstance(features, dict) else features.__dict__
        if is_training:
            return self._train_step(f, labels if isinstance(labels, dict) else labels.__dict__, params)
        source, source_length = f["source"], f["source_length"]
        target = None
        if is_validation:
            lab = labels if isinstance(labels, dict) else labels.__dict__
            target = lab["mel"]
        encoder_output = self.engine.encode(source, source_length)          # Embedding + EncoderV2
        mel_output, stop_token, decoder_state = self.decoder(encoder_output, is_training=False, is_validation=is_validation,
                                                             memory_sequence_length=source_length, target=target)
        alignment = decoder_state[0].alignment_history.stack().permute(1, 2, 0)     # models.py:53
        mel_postnet = self.postnet(mel_output)
        if is_validation:
            eng = self.engine
            mel_t, done, m1, m2 = (eng._f32(lab[k]) for k in ("mel", "done", "spec_loss_mask", "binary_loss_mask"))
            mel_loss = spec_loss(mel_output, mel_t, m1, engine=eng)
            done_loss = binary_loss(stop_token, done, m2, engine=eng)
            mel_tf, stop_tf, _ = self.decoder(encoder_output, is_training=False, is_validation=True,
                                              memory_sequence_length=source_length, target=target, teacher_forcing=True)
            mel_loss_tf = spec_loss(mel_tf, mel_t, m1, engine=eng)
            done_loss_tf = binary_loss(stop_tf, done, m2, engine=eng)
            return {"loss": mel_loss + done_loss, "eval_metric_ops": {
                "mel_loss": mel_loss, "done_loss": done_loss, "loss_with_teacher": mel_loss_tf + done_loss_tf,
                "mel_loss_with_teacher": mel_loss_tf, "done_loss_with_teacher": done_loss_tf},
                "mel": mel_output, "alignment": alignment}
        return {"predictions": {"id": f.get("id"), "mel": mel_output, "alignment": alignment, "source": source,
                                "text": f.get("text"), "mel_postnet": mel_postnet, "stop_token": stop_token}}

    def predict(self, features):
        return self.model_fn(features, None, ModeKeys.PREDICT)["predictions"]

    def close(self):
        for tr in self._trainers.values():
            tr.close()
        self._trainers = {}
        self.engine.close()
