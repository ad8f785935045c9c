"""Mirror of the reference's tacotron/rnn_impl.py: the "which LSTM kernel" switch.

In the reference the enum picks between tf.nn.rnn_cell.LSTMCell and LSTMBlockCell -- same math,
different TF kernels (rnn_impl.py:25-33).  Here both values select the one HIP implementation
(lstm_step_kernel in csrc/kernels.h); unknown values raise ValueError exactly as the reference does.
"""
from enum import Enum


class LSTMImpl(Enum):
    LSTMCell = "tf.nn.rnn_cell.LSTMCell"
    LSTMBlockCell = "tf.contrib.rnn.LSTMBlockCell"

    def __str__(self):
        return self.value

    @staticmethod
    def all_list():
        return list([str(i) for i in LSTMImpl])


class GRUImpl(Enum):
    GRUCell = "tf.contrib.rnn.GRUCell"
    GRUBlockCellV2 = "tf.contrib.rnn.GRUBlockCellV2"

    def __str__(self):
        return self.value

    @staticmethod
    def all_list():
        return list([str(i) for i in GRUImpl])


class LSTMCellSpec:
    """What lstm_cell_factory returns here: the description of one LSTM layer (units only);
    the arithmetic lives in the HIP library."""

    def __init__(self, num_units):
        self.num_units = num_units

    @property
    def state_size(self):
        return (self.num_units, self.num_units)      # LSTMStateTuple(c, h)

    @property
    def output_size(self):
        return self.num_units


def lstm_cell_factory(lstm_impl, num_units, dtype=None):
    """rnn_impl.py:25-33."""
    if isinstance(lstm_impl, str):
        lstm_impl = next((i for i in LSTMImpl if str(i) == lstm_impl), lstm_impl)
    if lstm_impl in (LSTMImpl.LSTMCell, LSTMImpl.LSTMBlockCell):
        return LSTMCellSpec(num_units)
    raise ValueError(f"Unknown LSTM cell implementation: {lstm_impl}. Supported: {', '.join(LSTMImpl.all_list())}")


class GRUCellSpec:
    """What gru_cell_factory returns here: the description of one TF GRUCell layer (reset gate applied before the candidate
    matmul); the arithmetic lives in the HIP library (gru_step_kernel / gru_seq_kernel, csrc/kernels.h)
    and is reached through tacotron_v1.py's assemblies."""

    def __init__(self, num_units):
        self.num_units = num_units

    @property
    def state_size(self):
        return self.num_units

    @property
    def output_size(self):
        return self.num_units


def gru_cell_factory(gru_impl, num_units, dtype=None):
    """rnn_impl.py:36-44: both names select the one HIP implementation; unknown ones raise ValueError like the reference."""
    if isinstance(gru_impl, str):
        gru_impl = next((i for i in GRUImpl if str(i) == gru_impl), gru_impl)
    if gru_impl in (GRUImpl.GRUCell, GRUImpl.GRUBlockCellV2):
        return GRUCellSpec(num_units)
    raise ValueError(f"Unknown GRU cell implementation: {gru_impl}. Supported: {', '.join(GRUImpl.all_list())}")
