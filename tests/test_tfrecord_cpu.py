"""TFRecord / tf.train.Example codec (tacotron2_amd/tfrecord.py; reference util/tfrecord.py) pinned by the CRC-32C
standard check value, by google.protobuf's own serializer on dynamically declared Example messages, and round trips."""
import numpy as np
import pytest

from tacotron2_amd import datasets as DS
from tacotron2_amd import tfrecord as R


def test_crc32c_known_answers():
    assert R.crc32c(b"123456789") == 0xE3069283            # the CRC-32C (Castagnoli) check value
    assert R.crc32c(b"") == 0
    assert R.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert R.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of ones
    assert R.masked_crc32c(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _example_classes():
    """tf.train.Example declared from scratch with google.protobuf (feature.proto / example.proto field numbers)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="t2_example_test.proto", package="t2test", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto
    def msg(name, *fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname, kw in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
            if kw.get("packed") is not None:
                f.options.packed = kw["packed"]
        return m
    msg("BytesList", ("value", 1, T.TYPE_BYTES, T.LABEL_REPEATED, None, {}))
    msg("FloatList", ("value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, None, {"packed": True}))
    msg("Int64List", ("value", 1, T.TYPE_INT64, T.LABEL_REPEATED, None, {"packed": True}))
    feat = msg("Feature", ("bytes_list", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t2test.BytesList", {}),
               ("float_list", 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t2test.FloatList", {}),
               ("int64_list", 3, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t2test.Int64List", {}))
    feat.oneof_decl.add(name="kind")
    for f in feat.field:
        f.oneof_index = 0
    feats = msg("Features", ("feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, ".t2test.Features.FeatureEntry", {}))
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    entry.field.add(name="value", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".t2test.Feature")
    msg("Example", ("features", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t2test.Features", {}))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        factory = message_factory.MessageFactory(pool)
        get = factory.GetPrototype
    return get(pool.FindMessageTypeByName("t2test.Example"))


def test_example_codec_against_google_protobuf():
    Example = _example_classes()
    mel = np.arange(12, dtype=np.float32).reshape(4, 3)
    feats = {"id": np.array([-7], np.int64), "mel": [mel.tobytes()], "lens": np.array([3, 1 << 40, 0], np.int64),
             "floats": np.array([0.5, -2.0], np.float32), "text": ["a".encode(), "ü".encode("utf-8")]}
    mine = R.encode_example(feats)
    ex = Example()
    ex.ParseFromString(mine)                                # google.protobuf reads what this codec wrote
    assert list(ex.features.feature["id"].int64_list.value) == [-7]
    assert list(ex.features.feature["lens"].int64_list.value) == [3, 1 << 40, 0]
    assert list(ex.features.feature["floats"].float_list.value) == [0.5, -2.0]
    assert list(ex.features.feature["text"].bytes_list.value) == feats["text"]
    assert ex.features.feature["mel"].bytes_list.value[0] == mel.tobytes()
    theirs = ex.SerializeToString(deterministic=True)       # and this codec reads what google.protobuf writes
    back = R.decode_example(theirs)
    assert set(back) == set(feats)
    np.testing.assert_array_equal(back["lens"], feats["lens"])
    np.testing.assert_array_equal(back["floats"], feats["floats"])
    assert back["text"] == feats["text"] and back["mel"] == feats["mel"]
    assert theirs == mine                                   # same bytes (sorted map keys, packed lists)


def test_reference_examples_round_trip_and_feed_the_batcher(tmp_path):
    rng = np.random.default_rng(0)
    srcs, tgts = [], []
    for i, (n_in, n_out) in enumerate(((5, 7), (9, 12), (4, 6))):
        ids = np.concatenate([rng.integers(32, 91, n_in - 1), [1]])
        spec, mel = rng.uniform(0, 1, (n_out, 6)).astype(np.float32), rng.uniform(0, 1, (n_out, 3)).astype(np.float32)
        sf, tf_ = str(tmp_path / f"s{i}.tfrecord"), str(tmp_path / f"t{i}.tfrecord")
        R.write_preprocessed_source_data2(i, "TEXT %d" % i, ids, "text2", ids[::-1], sf)
        R.write_preprocessed_target_data(i, spec, mel, tf_)
        (s_rec,), (t_rec,) = list(R.read_records(sf)), list(R.read_records(tf_))
        s, t = R.decode_preprocessed_source_data(s_rec), R.decode_preprocessed_target_data(t_rec)
        assert s["id"] == t["id"] == i and s["text"] == "TEXT %d" % i and s["source_length"] == n_in and s["source_length2"] == n_in
        np.testing.assert_array_equal(s["source"], ids)
        np.testing.assert_array_equal(s["source2"], ids[::-1])
        np.testing.assert_array_equal(t["spec"], spec)
        np.testing.assert_array_equal(t["mel"], mel)
        assert (t["target_length"], t["spec_width"], t["mel_width"]) == (n_out, 6, 3)
        srcs.append(sf); tgts.append(tf_)
    batches = list(DS.group_by_batch(R.read_examples(srcs, tgts), batch_size=4, approx_min_target_length=0))
    assert len(batches) == 1
    f, l = batches[0]
    assert f["source"].shape == (3, 9) and l["mel"].shape == (3, 12, 3) and l["target_length"].tolist() == [8, 12, 6]
    # a flipped byte is caught by the record checksum
    raw = bytearray(open(tgts[0], "rb").read())
    raw[20] ^= 1
    open(tgts[0], "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc"):
        list(R.read_records(tgts[0]))
