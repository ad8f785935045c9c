"""Dimension profiles for the Tacotron2 mel-generation path.

The reference keeps almost no V2 sizes in ``hparams.py`` (only ``attention_kernel``,
``attention_filters`` and ``cumulative_weights``, /root/reference/hparams.py:49-52); every other
V2 size is a constructor argument of the blocks in ``tacotron/tacotron_v2.py``.  ``T2Dims``
collects all of them; every field is a *runtime* parameter of the C-ABI (``t2_dims`` in
``include/t2mi.h``).

Profile ``P`` (paper sizes) is the benchmark default; profile ``R`` keeps the
reference's V1-flavoured hparams for cheap tests.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict, replace
from typing import Tuple


@dataclass(frozen=True)
class T2Dims:
    # reference hparams.py:14,23,27,28,50-52,87 and helpers.py:54
    num_symbols: int = 256
    num_mels: int = 80
    outputs_per_step: int = 2
    n_feed_frame: int = 2
    attention_kernel: int = 31
    attention_filters: int = 32
    cumulative_weights: bool = True
    max_iters: int = 500
    min_iters: int = 10
    # constructor arguments of tacotron_v2.py blocks (no hparams entry in the reference)
    embedding_dim: int = 512
    enc_conv_layers: int = 3
    enc_kernel: int = 5
    enc_units: int = 512            # E (BiLSTM: E/2 per direction), tacotron_v2.py:40
    prenet_units: Tuple[int, ...] = (256, 256)
    attention_lstm_units: int = 1024   # A
    decoder_lstm_units: int = 1024     # D
    attention_dim: int = 128           # U
    postnet_layers: int = 5
    postnet_kernel: int = 5
    postnet_channels: int = 512
    zoneout_cell: float = 0.1
    zoneout_output: float = 0.1
    bn_eps: float = 1e-3               # tf.layers.batch_normalization default
    forget_bias: float = 1.0           # tf.nn.rnn_cell.LSTMCell default

    def __post_init__(self):
        # same checks the reference makes (tacotron_v2.py:40, helpers.py:55, modules.py:241-245)
        assert self.enc_units % 2 == 0
        assert self.n_feed_frame <= self.outputs_per_step
        zm = min(self.zoneout_output, self.zoneout_cell)
        zs = max(self.zoneout_output, self.zoneout_cell)
        if zm < 0. or zs > 1.:
            raise ValueError('One/both provided Zoneout factors are not in [0, 1]')

    def replace(self, **kw) -> "T2Dims":
        return replace(self, **kw)

    def asdict(self):
        return asdict(self)


#: Tacotron2-paper sizes (SURVEY.md §8d "profile P") -- benchmark default.
PROFILE_P = T2Dims()

#: reference V1-flavoured hparams (hparams.py:31,45-47) -- cheap tests.
PROFILE_R = T2Dims(embedding_dim=256, enc_units=256, prenet_units=(256, 128),
                   attention_lstm_units=256, decoder_lstm_units=256, attention_dim=256,
                   postnet_channels=256)

#: tiny sizes, deliberately not multiples of the kernel tile sizes.
PROFILE_TINY = T2Dims(num_symbols=40, num_mels=5, outputs_per_step=2, n_feed_frame=1,
                      attention_kernel=5, attention_filters=3, embedding_dim=6,
                      enc_conv_layers=2, enc_kernel=3, enc_units=10, prenet_units=(7, 9),
                      attention_lstm_units=11, decoder_lstm_units=13, attention_dim=6,
                      postnet_layers=3, postnet_kernel=3, postnet_channels=7,
                      zoneout_cell=0.1, zoneout_output=0.2, max_iters=12, min_iters=3)


@dataclass(frozen=True)
class V1Dims:
    """Tacotron v1 sizes: every field is an entry of the reference's hparams.py (:14-15,23,27-28,31-47,
    54-60,87) with the reference's default value (BASELINE.json configs[4])."""
    num_symbols: int = 256
    num_mels: int = 80
    num_freq: int = 1025
    outputs_per_step: int = 2
    n_feed_frame: int = 2
    max_iters: int = 500
    min_iters: int = 10
    embedding_dim: int = 256
    # EncoderV1 (tacotron_v1.py:20-39)
    encoder_prenet_out_units: Tuple[int, int] = (256, 128)
    cbhg_out_units: int = 256
    conv_channels: int = 128
    max_filter_width: int = 16
    projection1_out_channels: int = 128
    projection2_out_channels: int = 128
    num_highway: int = 4
    # DecoderV1 (tacotron_v1.py:96-103)
    decoder_prenet_out_units: Tuple[int, int] = (256, 128)
    attention_out_units: int = 256
    decoder_out_units: int = 256
    # PostNet (tacotron_v1.py:158-164)
    post_net_cbhg_out_units: int = 256
    post_net_conv_channels: int = 128
    post_net_max_filter_width: int = 8
    post_net_projection1_out_channels: int = 256
    post_net_projection2_out_channels: int = 80
    post_net_num_highway: int = 4
    bn_eps: float = 1e-3

    def __post_init__(self):
        assert self.n_feed_frame <= self.outputs_per_step                        # helpers.py:55
        assert self.cbhg_out_units % 2 == 0 and self.post_net_cbhg_out_units % 2 == 0      # modules.py:169
        assert self.encoder_prenet_out_units[0] == self.embedding_dim            # tacotron_v1.py:41-44
        # shape constraints the TF graph enforces implicitly: CBHG residual (modules.py:214) and
        # ResidualWrapper (tacotron_v1.py:72-73)
        assert self.projection2_out_channels == self.encoder_prenet_out_units[-1]
        assert self.post_net_projection2_out_channels == self.num_mels

    def replace(self, **kw) -> "V1Dims":
        return replace(self, **kw)


#: the reference's hparams.py defaults
V1_PROFILE = V1Dims()

#: tiny sizes, deliberately not multiples of the kernel tile sizes (even and odd bank widths)
V1_TINY = V1Dims(num_symbols=30, num_mels=5, num_freq=9, outputs_per_step=2, n_feed_frame=1, max_iters=10, min_iters=3,
                 embedding_dim=6, encoder_prenet_out_units=(6, 5), cbhg_out_units=8, conv_channels=3, max_filter_width=4,
                 projection1_out_channels=7, projection2_out_channels=5, num_highway=2,
                 decoder_prenet_out_units=(7, 6), attention_out_units=9, decoder_out_units=11,
                 post_net_cbhg_out_units=6, post_net_conv_channels=3, post_net_max_filter_width=3,
                 post_net_projection1_out_channels=4, post_net_projection2_out_channels=5, post_net_num_highway=2)
