"""Weight containers and TensorFlow-variable import (tacotron2_amd/checkpoint.py).  CPU only."""
import numpy as np
import pytest

from tacotron2_amd import checkpoint as C
from tacotron2_amd.hparams import PROFILE_R, PROFILE_TINY, T2Dims
from tacotron2_amd.weights import init_weights, weight_shapes

MID = T2Dims(num_symbols=64, num_mels=20, outputs_per_step=2, n_feed_frame=2, embedding_dim=64, enc_units=64,
             prenet_units=(48, 32), attention_lstm_units=96, decoder_lstm_units=80, attention_dim=32, postnet_channels=64)


def test_npz_round_trip(tmp_path):
    w = init_weights(MID, 3, perturb=True)
    p = str(tmp_path / "w.npz")
    C.save_npz(p, w)
    got = C.load_npz(p, MID)
    assert set(got) == set(w)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k])
    with pytest.raises((KeyError, ValueError)):
        C.load_npz(p, PROFILE_TINY)                      # another model: names / shapes do not fit


@pytest.mark.parametrize("d", [MID, PROFILE_R, PROFILE_TINY])
def test_import_tf_variables(d):
    w = init_weights(d, 5, perturb=True)
    names = C.tf_style_names(d)
    assert set(names) == set(weight_shapes(d))
    rng = np.random.default_rng(0)
    tf_vars = {names[k] + ":0": v for k, v in w.items()}
    tf_vars["global_step"] = np.array(7, np.int64)       # ignored, as are the optimizer slots
    tf_vars[names["embedding"] + "/Adam"] = np.zeros_like(w["embedding"])
    tf_vars[names["embedding"] + "/Adam_1"] = np.zeros_like(w["embedding"])
    tf_vars["beta1_power"] = np.array(0.9, np.float32)
    keys = list(tf_vars)
    rng.shuffle(keys)
    got = C.import_tf_variables({k: tf_vars[k] for k in keys}, d)
    assert set(got) == set(w)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k], err_msg=k)


def test_import_errors_and_name_map():
    d = MID
    w = init_weights(d, 6)
    names = C.tf_style_names(d)
    tf_vars = {names[k]: v for k, v in w.items()}
    missing = dict(tf_vars)
    del missing[names["attention/attention_bias"]]
    with pytest.raises(KeyError, match="attention_bias"):
        C.import_tf_variables(missing, d)
    bad = dict(tf_vars)
    bad[names["embedding"]] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError, match="shape"):
        C.import_tf_variables(bad, d)
    dup = dict(tf_vars)
    dup["other_scope/embedding"] = w["embedding"]
    with pytest.raises(ValueError, match="claimed"):
        C.import_tf_variables(dup, d)
    # a name the rules cannot place goes through name_map
    odd = dict(tf_vars)
    odd["my_scope/stop_w"] = odd.pop(names["decoder/stop_token_projection/kernel"])[:, :]
    odd["my_scope/stop_w"] = odd["my_scope/stop_w"].reshape(d.decoder_lstm_units, 1)
    got = C.import_tf_variables(odd, d, name_map={"my_scope/stop_w": "decoder/stop_token_projection/kernel"})
    np.testing.assert_array_equal(got["decoder/stop_token_projection/kernel"], w["decoder/stop_token_projection/kernel"])
