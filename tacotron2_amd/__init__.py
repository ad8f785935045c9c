"""tacotron2_amd -- MI355X (gfx950) native Tacotron2 mel generation behind the class / call API of
nii-yamagishilab/tacotron2 (tacotron_v2.py blocks, rnn_wrappers.py cells, helpers.py helpers,
a synthesize driver).  Arithmetic: hand-written HIP kernels in libt2mi.so (include/t2mi.h)."""
from .hparams import T2Dims, PROFILE_P, PROFILE_R, PROFILE_TINY  # noqa: F401
from .weights import init_weights, weight_shapes  # noqa: F401
