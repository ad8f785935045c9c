"""world_size-2 gloo rehearsal of the N>1 path of bench.py: ranks take disjoint shards, no data-path
collective; only the timing reduction (MAX of seconds, SUM of frames) crosses ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tacotron2_amd.sharding import make_batches, shard_batches
    lens = np.random.default_rng(3).integers(60, 141, 256)
    batches = make_batches(lens, 64)
    mine = shard_batches(len(batches), world, rank)
    frames = float(sum(len(batches[b]) * 800 for b in mine))
    tot = torch.tensor([0.5 + rank, frames], dtype=torch.float64)
    tmax = tot.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    q.put((rank, mine, float(tmax[0]), float(tot[1])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]
    assert res[0][2] == res[1][2] == 1.5 and res[0][3] == res[1][3] == 256 * 800.0


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tacotron2_amd import training as TR
    from oracle import t2_oracle as O
    rng = np.random.default_rng(100 + rank)
    g = torch.tensor(rng.standard_normal(1024).astype(np.float32))
    scale = TR.allreduce_gradients(g)                       # in-place sum; mean factor handed to the optimizer
    # per-rank loss parts -> the loss of the global batch (SUM_BY_NONZERO_WEIGHTS is not a mean of means)
    B = 2 + rank
    l = rng.uniform(0, 1, (B, 5, 3)).astype(np.float32)
    mask = (np.arange(5)[None, :] < np.array([5, 2, 4, 1][:B])[:, None]).astype(np.float32)
    num, cnt = O.loss_parts(l, mask[:, :, None])
    loss = TR.allreduce_loss_parts(torch.tensor([num, cnt, 0.0], dtype=torch.float32))
    q.put((rank, g.numpy(), scale, float(loss), l, mask))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_and_loss_exchange():
    """The training-step exchange of SURVEY.md 8e on gloo: one sum all-reduce of the flat gradient buffer
    (+ 1/world handed to the optimizer) and the (numerator, count) pairs of the losses."""
    from oracle import t2_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0 = np.random.default_rng(100).standard_normal(1024).astype(np.float32)
    g1 = np.random.default_rng(101).standard_normal(1024).astype(np.float32)
    for r in res:
        np.testing.assert_allclose(r[1], g0 + g1, rtol=1e-6)
        assert r[2] == 0.5
    l_all = np.concatenate([res[0][4], res[1][4]])
    m_all = np.concatenate([res[0][5], res[1][5]])
    num, cnt = O.loss_parts(l_all, m_all[:, :, None])
    assert abs(res[0][3] - num / cnt) < 1e-6 and res[0][3] == res[1][3]


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import torch_train as TT
    from tacotron2_amd import training as TR
    d, w, ids, lens, labels = _dp_case()
    rows = np.arange(3 * rank, 3 * rank + 3)
    lab = {k: v[rows] for k, v in labels.items()}
    counts = TR.global_loss_counts(lab["spec_loss_mask"], lab["binary_loss_mask"], d.num_mels)      # sum all-reduce over the ranks
    ref = TT.train_forward_backward(d, w, ids[rows], lens[rows], lab, seed=5, prenet_drop_rate=0.0, encoder_drop_rate=0.0,
                                    loss_counts=(float(counts[0]), float(counts[1])))
    names = sorted(ref["grads"])
    g = torch.tensor(np.concatenate([ref["grads"][k].reshape(-1) for k in names]))
    # the step's gradient all-reduce, plain SUM, in the two buckets Trainer.train_step uses: the attention/ + decoder/ prefix
    # (final after the decoder backward; started asynchronously, it runs beside the encoder backward) and the rest
    spec = TR.flatten_spec({k: ref["grads"][k] for k in names})
    flat = torch.tensor(TR.flatten({k: ref["grads"][k] for k in names}, spec))
    cut = TR.decoder_prefix_end(spec)
    assert 0 < cut < flat.numel() and spec[0][0].startswith("attention/")
    works = [TR.allreduce_sum_async(flat[:cut]), TR.allreduce_sum_async(flat[cut:])]
    for wk in works:
        wk.wait()
    g = torch.cat([flat[off:off + size] for _, _, off, size in spec])
    loss = TR.allreduce_sum(torch.tensor([ref["loss"]], dtype=torch.float64))
    q.put((rank, g.numpy(), float(loss[0]), counts.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _dp_case():
    from tacotron2_amd.hparams import PROFILE_TINY
    from tacotron2_amd.weights import init_weights
    d = PROFILE_TINY.replace(enc_conv_layers=0, embedding_dim=10, zoneout_cell=0.0, zoneout_output=0.0)     # nothing but the loss couples the rows
    w = init_weights(d, 2, perturb=True)
    rng = np.random.default_rng(9)
    B, T, S, r = 6, 7, 4, d.outputs_per_step
    ids = rng.integers(0, d.num_symbols, (B, T))
    lens = np.array([T, 4, 6, T, 3, 5], np.int64)
    tl = np.array([S * r, 3, 5, S * r, 2, 7])
    sl = (tl + r - 1) // r
    labels = {"mel": rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32),
              "done": (np.arange(S)[None, :] >= (sl[:, None] - 1)).astype(np.float32),
              "spec_loss_mask": (np.arange(S * r)[None, :] < tl[:, None]).astype(np.float32),
              "binary_loss_mask": (np.arange(S)[None, :] < sl[:, None]).astype(np.float32)}
    return d, w, ids, lens, labels


def test_two_rank_training_step_is_loss_exact():
    """two ranks x 3 rows = one batch of 6 rows under SUM_BY_NONZERO_WEIGHTS (models.py:136-148): the ranks sum
    their non-zero weight counts, each differentiates sum_rank / count_total, and the plain SUM of the gradients (and of the
    loss shares) equals the single-process gradient (loss) of the whole batch -- on the oracle, over gloo."""
    from oracle import torch_train as TT
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, w, ids, lens, labels = _dp_case()
    full = TT.train_forward_backward(d, w, ids, lens, labels, seed=5, prenet_drop_rate=0.0, encoder_drop_rate=0.0)
    names = sorted(full["grads"])
    g_full = np.concatenate([full["grads"][k].reshape(-1) for k in names])
    for r in res:
        np.testing.assert_allclose(r[1], g_full, atol=1e-6 * float(np.abs(g_full).max()) + 1e-9)
        assert abs(r[2] - full["loss"]) < 1e-6
        assert r[3][0] == float((labels["spec_loss_mask"] != 0).sum() * d.num_mels) and r[3][1] == float((labels["binary_loss_mask"] != 0).sum())


def test_rank_seed_decorrelates_the_ranks_masks():
    """the counter-based masks are keyed on the LOCAL row, so data-parallel ranks must not share the seed.
    rank_seed is the identity on one device (single-device results stay reproducible from the caller's seed) and gives every
    rank its own 32-bit seed otherwise; the oracle's masks drawn from those seeds are uncorrelated across ranks."""
    from oracle import t2_oracle as O
    from tacotron2_amd import training as TR
    assert TR.rank_seed(1234, 0, 1) == 1234 and TR.rank_seed(-1, 0, 1) == 0xFFFFFFFF
    seeds = {TR.rank_seed(7, r, 8) for r in range(8)}
    assert len(seeds) == 8 and 7 not in seeds and all(0 <= s < 2 ** 32 for s in seeds)
    assert TR.rank_seed(7, 3, 8) == TR.rank_seed(7, 3, 8) != TR.rank_seed(8, 3, 8)
    u0 = O.rng_u01(TR.rank_seed(7, 0, 2), O.STREAM_PRENET, 0, 32, 256) >= 0.5
    u1 = O.rng_u01(TR.rank_seed(7, 1, 2), O.STREAM_PRENET, 0, 32, 256) >= 0.5
    agree = float((u0 == u1).mean())
    assert 0.45 < agree < 0.55, agree                      # identical seeds would give 1.0
