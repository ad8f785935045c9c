"""Tensor contract of the reference's input pipeline (tacotron2_amd/datasets.py; reference datasets/dataset.py)."""
import numpy as np
import pytest

from tacotron2_amd import datasets as DS


def test_prepare_target_padding_rule():
    rng = np.random.default_rng(0)
    mel = rng.uniform(0, 1, (7, 4)).astype(np.float32)
    t = DS.prepare_target(mel, r=2)                       # 7 % 2 != 0 -> (7 // 2 + 1) * 2 = 8   (dataset.py:113-131)
    assert t["target_length"] == 8 and t["mel"].shape == (8, 4)
    np.testing.assert_array_equal(t["mel"][:7], mel)
    assert not t["mel"][7].any()
    np.testing.assert_array_equal(t["done"], [0, 0, 0, 1])             # zeros(8/2 - 1) ++ [1]   (:133-135)
    np.testing.assert_array_equal(t["spec_loss_mask"], np.ones(8))      # the tail padding is NOT masked (:138)
    np.testing.assert_array_equal(t["binary_loss_mask"], np.ones(4))
    t = DS.prepare_target(mel[:6], r=2)                   # already a multiple: no padding (not 6 // 2 + 1 groups)
    assert t["target_length"] == 6 and t["done"].shape == (3,)
    t = DS.prepare_target(mel, r=3, spec=np.ones((7, 5), np.float32))
    assert t["target_length"] == 9 and t["spec"].shape == (9, 5) and not t["spec"][7:].any()
    with pytest.raises(ValueError):
        DS.prepare_target(mel, r=2, spec=np.ones((6, 5), np.float32))


def test_bucket_key_keeps_the_reference_quirk():
    # tf.minimum(len - 100, 0) // 50: every utterance of >= 100 frames shares bucket 0 (dataset.py:213-216)
    assert DS.bucket_key(100) == 0 and DS.bucket_key(800) == 0 and DS.bucket_key(100000) == 0
    assert DS.bucket_key(99) == -1 and DS.bucket_key(50) == -1 and DS.bucket_key(49) == -2 and DS.bucket_key(0) == -2


def test_filter_sees_the_padded_length():
    assert DS.keep_by_max_output_length(1000) and not DS.keep_by_max_output_length(1001)
    src = DS.prepare_source([5, 6, 1])
    batches = list(DS.group_by_batch([(src, np.zeros((999, 3), np.float32)), (src, np.zeros((1001, 3), np.float32))],
                                     batch_size=4))
    assert len(batches) == 1 and batches[0][1]["target_length"].tolist() == [1000]     # 999 -> 1000 kept, 1001 -> 1002 dropped


def test_padded_batch_padding_values():
    rng = np.random.default_rng(1)
    srcs = [DS.prepare_source(rng.integers(32, 91, n), example_id=i) for i, n in enumerate((5, 3, 4))]
    tgts = [DS.prepare_target(rng.uniform(0, 1, (n, 2)).astype(np.float32), r=2, example_id=i) for i, n in enumerate((6, 3, 9))]
    f, l = DS.padded_batch(srcs, tgts)
    assert f["source"].shape == (3, 5) and f["source"][1, 3:].tolist() == [0, 0] and f["source_length"].tolist() == [5, 3, 4]
    assert l["mel"].shape == (3, 10, 2) and l["target_length"].tolist() == [6, 4, 10]
    np.testing.assert_array_equal(l["done"], [[0, 0, 1, 1, 1], [0, 1, 1, 1, 1], [0, 0, 0, 0, 1]])       # padded with 1 (:256)
    np.testing.assert_array_equal(l["binary_loss_mask"], [[1, 1, 1, 0, 0], [1, 1, 0, 0, 0], [1, 1, 1, 1, 1]])
    np.testing.assert_array_equal(l["spec_loss_mask"][1], [1, 1, 1, 1, 0, 0, 0, 0, 0, 0])
    assert not l["mel"][0, 6:].any()


def test_group_by_window_order_and_sizes():
    rng = np.random.default_rng(2)
    lens = [120] * 11 + [60] * 3 + [130] * 2          # keys: 0 x11, -1 x3, 0 x2
    ex = [(DS.prepare_source([7, 1], example_id=i), rng.uniform(0, 1, (n, 2)).astype(np.float32)) for i, n in enumerate(lens)]
    out = list(DS.group_by_batch(ex, batch_size=2))   # window = 10
    ids = [b[0]["id"].tolist() for b in out]
    # the first full window of key 0 (ids 0..9) is emitted as 5 batches of 2 as soon as it fills; the rest at the end,
    # key 0's partial window (10, 14, 15) before key -1's (11, 12, 13) because key 0 appeared first
    assert ids == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 14], [15], [11, 12], [13]]
