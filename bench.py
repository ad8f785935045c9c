#!/usr/bin/env python3
"""bench.py -- mel frames/sec of Tacotron2 free-running synthesis, B=64 per GPU (BASELINE.json
config 3), on N GPUs of one node (utterances batch-sharded, no data-path collective), plus the
batch-1 RTF (config 2) and a bounded CPU baseline (the NumPy oracle) at N=1.

One "step" = one pass of the hot path over one batch of 64 synthetic utterances per GPU:
Embedding + EncoderV2 -> free-running decode (stop rule evaluated on the device every step) ->
PostNetV2.  Weights are random-initialised at the Tacotron2-paper sizes (profile P);
the stop bias is set so that the rule never fires and every decode runs to max_iters = 400 steps
(800 mel frames per utterance): the amount of work is fixed and stated.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md, HBM3E spec
FRAME_SHIFT_S = 0.0125             # reference hparams.py:18


def make_inputs(B, t_lo, t_hi, seed):
    """BASELINE.json config 3: T_in ~ uniform{t_lo..t_hi}, ids upper-case ASCII + eos=1
    (reference datasets/ljspeech.py:154-156, datasets/corpus.py:80), padded per shard."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(t_lo, t_hi + 1, size=B)
    T = int(lens.max())
    ids = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lens):
        ids[b, :n - 1] = rng.integers(32, 91, size=n - 1)
        ids[b, n - 1] = 1
    return ids, lens.astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--max-iters", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch1", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (tuning runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    # T2MI_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices; timing reductions on the CPU).  The driver's runs use the default: one rank per GPU, RCCL.
    backend = os.environ.get("T2MI_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from tacotron2_amd.engine import Engine
    from tacotron2_amd.hparams import PROFILE_P
    from tacotron2_amd.weights import init_weights

    d = PROFILE_P.replace(max_iters=args.max_iters)
    w = init_weights(d, seed=1234)                       # same weights on every GPU
    w["decoder/stop_token_projection/bias"] = np.array([-1e3], np.float32)   # decode runs to the cap
    eng = Engine(d, w, device=local_rank)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    B, S, r, M = args.batch, args.max_iters, d.outputs_per_step, d.num_mels
    # The job = world x B utterances, T_in ~ U{60..140}; the product's sharding code (tacotron2_amd/sharding.py, used by
    # synthesize.synthesize_sharded) sorts them by length, cuts "reference batches" of B rows and deals batch b to rank
    # b % world: every rank gets exactly one batch here, no data-path collective.
    from tacotron2_amd.sharding import make_batches, shard_batches
    ids_all, lens_all = make_inputs(B * world, 60, 140, seed=1234 + 3)
    batches = make_batches(lens_all, B)
    mine = shard_batches(len(batches), world, rank)
    assert len(mine) == 1
    rows = batches[mine[0]]
    lens = lens_all[rows]
    ids = np.ascontiguousarray(ids_all[rows][:, :int(lens.max())])
    T_in = ids.shape[1]
    bufs = eng.decode_free_buffers(B, T_in, S)

    def synth():
        memory = eng.encode(ids, lens)
        eng.decode_free_launch(memory, lens, S, d.min_iters, bufs)
        n = eng.decode_free_finish()
        mel = bufs[0][:, :n * r]
        post = eng.postnet(mel)
        return n, post

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        synth()
    barrier()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        n, _post = synth()
        frames += B * n * r
    barrier()
    dt = time.perf_counter() - t0
    tot = torch.tensor([dt, float(frames)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, frames_all = float(tmax[0]), float(tot[1])
    else:
        dt_max, frames_all = dt, float(frames)

    out = None
    parity_failed = False
    if rank == 0:
        out = {
            "metric": "mel frames/sec (node), Tacotron2 decode B=64 at 1/2/4/8 GPUs; batch-1 RTF",
            "value": frames_all / dt_max, "unit": "mel frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: Tacotron2 free-running synthesis, B=64 per GPU, "
                                   "profile P (paper sizes), T_in~U{60..140}, decode to max_iters=%d steps (r=2 -> %d "
                                   "frames/utterance), encoder + decoder + PostNetV2, random-init weights" % (S, S * r),
                       "batch_per_gpu": B, "T_in": int(T_in), "decoder_steps": S,
                       "parallelism": "batch-sharded x%d (utterances of the job sorted by length, dealt as batches of %d: "
                                      "tacotron2_amd/sharding.py), no data-path collective" % (world, B)},
        }

    # ---- roofline of the dominant kernel: pz::decode_kernel<4, 0> (csrc/persist.h), ONE launch = the whole free-running decode
    # of the batch (all S decoder steps).  Live HIP-event timing on the library's stream around the decode (the events also
    # bracket the per-utterance set-up -- memory mask, keys GEMM, state memset: < 1 % at S = 400).  Algorithmic work per launch
    # = S x the per-step figures of SURVEY.md 8d (Fd, Bd), both stated in DESIGN.md. ----
    if rank == 0:
        memory = eng.encode(ids, lens)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps, ms_list = 5, []
        for _ in range(reps):
            eng.decode_free_launch(memory, lens, S, d.min_iters, bufs, events=(e0, e1))
            n_dec = eng.decode_free_finish()
            torch.cuda.synchronize()
            ms_list.append(e0.elapsed_time(e1))
        ms_dec = float(np.mean(ms_list))
        persist_ran = eng.get_stat("persist_launches") > 0
        P2_, E_, A_, D_, U_ = d.prenet_units[-1], d.enc_units, d.attention_lstm_units, d.decoder_lstm_units, d.attention_dim
        wd = (M * d.n_feed_frame * d.prenet_units[0] + d.prenet_units[0] * P2_ + (P2_ + E_ + A_) * 4 * A_ + A_ * U_ +
              (A_ + E_ + D_) * 4 * D_ + 2 * D_ * 4 * D_ + D_ * (M * r + 1))
        step_flops = B * (2.0 * wd + T_in * (2 * d.attention_kernel * d.attention_filters + 2 * d.attention_filters * U_ + 6 * U_ + 2 * E_ + 8))
        step_bytes = 4.0 * (wd + d.attention_kernel * d.attention_filters + d.attention_filters * U_) + \
            B * 4.0 * (T_in * (U_ + E_) + 3 * T_in + 2 * 2 * (A_ + 2 * D_) + M * d.n_feed_frame + M * r + 1)
        flops_launch, bytes_launch = step_flops * n_dec, step_bytes * n_dec
        achieved = flops_launch / (ms_dec * 1e-3) / 1e12
        # `traffic` cannot be measured inside this process (PMC counters need rocprofv3 around it): it is the figure of the last
        # committed counter passes of THIS kernel (tools/evidence.sh -> tools/collect_profiles.py -> profiles/traffic.json), and
        # the line says which capture (commit, library hash) it quotes
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(tj.get("current", "r02_persistent_decode_b64"), {})
                traffic, traffic_source = ent.get("hbm_bytes_per_launch"), ent.get("source")
            except Exception:
                traffic = None
        # the launch-per-phase schedule of round 1 (6 launches per step, hipGraph chunks), same decode, for comparison
        eng.set_option("persist", 0)
        eng.decode_free_launch(memory, lens, S, d.min_iters, bufs)
        eng.decode_free_finish()
        eng.decode_free_launch(memory, lens, S, d.min_iters, bufs, events=(e0, e1))
        eng.decode_free_finish()
        torch.cuda.synchronize()
        ms_launches = e0.elapsed_time(e1)
        eng.set_option("persist", 1)
        for kv in args.opt:
            k, v = kv.split("=")
            eng.set_option(k, int(v))
        out["roofline"] = {"kernel": "pz::decode_kernel<4, 0> (persistent free-running decoder: one launch = %d decoder steps of the B=%d batch)" % (n_dec, B)
                                     if persist_ran else "launch-per-phase schedule (persistent decoder not eligible)",
                           "bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                           "avg_launch_ms": ms_dec, "launches_timed": reps, "us_per_decoder_step": ms_dec * 1e3 / n_dec,
                           "algorithmic_flops_per_launch": flops_launch, "algorithmic_bytes_per_launch": bytes_launch,
                           "algorithmic_flops_per_step": step_flops, "algorithmic_bytes_per_step": step_bytes,
                           "hbm_side": {"achieved_GBs": bytes_launch / (ms_dec * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
                                        "frac": bytes_launch / (ms_dec * 1e-3) / 1e9 / HBM_PEAK_GBS},
                           "launch_schedule_round1": {"us_per_decoder_step": ms_launches * 1e3 / n_dec,
                                                      "frac_of_fp32_mfma_peak": step_flops / (ms_launches * 1e-3 / n_dec) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                                      "note": "same decode with option persist=0: 6 launches per step replayed from hipGraph chunks"}}

    # ---- batch-1 latency / RTF (BASELINE.json configs[1]) ----
    if rank == 0 and not args.no_batch1:
        ids1, lens1 = make_inputs(1, 100, 100, seed=1235)
        b1 = eng.decode_free_buffers(1, 100, S)

        def synth1():
            mem = eng.encode(ids1, lens1)
            eng.decode_free_launch(mem, lens1, S, d.min_iters, b1)
            n = eng.decode_free_finish()
            eng.postnet(b1[0][:, :n * r])
            return n
        synth1()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            n1 = synth1()
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / reps
        # decode only, HIP events on the library's stream: the GEMV-bound part (per-step weights read once,
        # keys + values of the utterance, states) against the HBM roofline
        mem1 = eng.encode(ids1, lens1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.decode_free_launch(mem1, lens1, S, d.min_iters, b1, events=(ev0, ev1))
        nd = eng.decode_free_finish()
        torch.cuda.synchronize()
        us_step = ev0.elapsed_time(ev1) * 1e3 / nd
        P2_, E_, A_, D_, U_ = d.prenet_units[-1], d.enc_units, d.attention_lstm_units, d.decoder_lstm_units, d.attention_dim
        wd1 = (M * d.n_feed_frame * d.prenet_units[0] + d.prenet_units[0] * P2_ + (P2_ + E_ + A_) * 4 * A_ + A_ * U_ +
               d.attention_kernel * d.attention_filters + d.attention_filters * U_ + (A_ + E_ + D_) * 4 * D_ + 2 * D_ * 4 * D_ +
               D_ * (M * r + 1))
        bytes_step = 4.0 * wd1 + 4.0 * (100 * (U_ + E_) + 3 * 100 + 4 * (A_ + 2 * D_) + M * d.n_feed_frame + M * r + 1)
        out["batch1"] = {"frames": n1 * r, "ms": dt1 * 1e3, "frames_per_s": n1 * r / dt1,
                         "rtf": dt1 / (n1 * r * FRAME_SHIFT_S), "us_per_decoder_step_incl_enc_post": dt1 * 1e6 / n1,
                         "decode_only_us_per_step": us_step,
                         "hbm_roofline": {"algorithmic_bytes_per_step": bytes_step, "achieved_GBs": bytes_step / us_step / 1e3,
                                          "peak_GBs": HBM_PEAK_GBS, "frac": bytes_step / us_step / 1e3 / HBM_PEAK_GBS,
                                          "note": "persistent decoder: latency-bound by its 5 all-to-all hops + 3 group hops per step, not by the weight stream"}}

    # ---- teacher-forced decoder forward, B = 1, 100 characters / 800 frames (BASELINE.json configs[0], the reference's own
    # CPU-runnable case): encoder + teacher-forced decode + PostNetV2 on the GPU, the NumPy port on the host beside it ----
    if rank == 0 and not args.no_batch1:
        try:
            ids0, lens0 = make_inputs(1, 100, 100, seed=1234)
            tgt0 = torch.tensor(np.random.default_rng(1234).uniform(0, 1, (1, S * r, M)).astype(np.float32), device="cuda")

            def teacher1():
                mem = eng.encode(ids0, lens0)
                mel_t, stop_t, _ = eng.decode_teacher(mem, lens0, tgt0, teacher_forcing=True)
                return mel_t, eng.postnet(mel_t)
            teacher1()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(3):
                mel_t, _ = teacher1()
            torch.cuda.synchronize()
            dt0 = (time.perf_counter() - t0_) / 3
            out["teacher_forced_b1"] = {"workload": "BASELINE.json configs[0]: encoder + teacher-forced decoder forward + PostNetV2, B=1, "
                                                    "T_in=100, %d frames" % (S * r), "ms": dt0 * 1e3, "frames_per_s": S * r / dt0}
            if world == 1 and not args.no_cpu_baseline:
                from oracle import t2_oracle as O

                def cpu_once(steps):
                    tc = time.perf_counter()
                    mem_o = O.encoder_v2(d, w, O.embedding_lookup(ids0, w["embedding"]), lens0)
                    mel_o, _, _ = O.decoder_v2(d, w, mem_o, is_validation=True, teacher_forcing=True, memory_sequence_length=lens0,
                                               target=tgt0.cpu().numpy()[:, :steps * r], max_iters=steps)
                    O.postnet_v2(d, w, mel_o)
                    return time.perf_counter() - tc, mel_o
                # protocol of SURVEY.md 8d: warm-up, then the median of 3 runs on all host threads; the 1-thread figure on a
                # quarter of the steps (scaled), also the median of 3
                cpu_once(S // 8)
                runs = [cpu_once(S) for _ in range(3)]
                dtc = float(np.median([t_ for t_, _ in runs]))
                mel_o = runs[-1][1]
                one = None
                try:
                    from threadpoolctl import threadpool_limits
                    with threadpool_limits(limits=1):
                        one = float(np.median([cpu_once(S // 4)[0] for _ in range(3)])) * 4.0
                except Exception:
                    one = None
                out["teacher_forced_b1"]["cpu_port"] = {"ms": dtc * 1e3, "frames_per_s": S * r / dtc, "kind": "port", "runs": 3, "statistic": "median",
                                                        "one_thread_ms_scaled_from_quarter": None if one is None else one * 1e3,
                                                        "one_thread_frames_per_s": None if one is None else S * r / one,
                                                        "max_abs_mel_diff": float(np.abs(mel_t.cpu().numpy() - mel_o).max())}
        except Exception as ex:
            out["teacher_forced_b1"] = {"error": repr(ex)}

    # ---- training step (BASELINE.json configs[3]; N=1 only: tools/train_bench.py is the multi-rank version) ----
    if rank == 0 and world == 1 and not args.no_train:
        try:
            from tacotron2_amd import training as TR
            Bt, Tt, St = 32, 140, 400
            rng = np.random.default_rng(1234 + 4)
            lens_t = rng.integers(60, Tt + 1, Bt).astype(np.int64); lens_t[0] = Tt
            ids_t = np.zeros((Bt, Tt), np.int64)
            for b, n in enumerate(lens_t):
                ids_t[b, :n] = rng.integers(32, 91, n)
            tl = rng.integers(400, 801, Bt); tl[0] = 800
            sl = (tl + 1) // 2
            labels = {"mel": rng.uniform(0, 1, (Bt, St * r, M)).astype(np.float32),
                      "done": (np.arange(St)[None, :] >= (sl[:, None] - 1)).astype(np.float32),
                      "spec_loss_mask": (np.arange(St * r)[None, :] < tl[:, None]).astype(np.float32),
                      "binary_loss_mask": (np.arange(St)[None, :] < sl[:, None]).astype(np.float32)}
            labels = {k: torch.tensor(v, device="cuda") for k, v in labels.items()}
            feats = {"source": ids_t, "source_length": lens_t}
            spec = TR.flatten_spec(w)
            trn = TR.Trainer(eng, Bt, Tt, St)
            trn.bind_gradients(spec)
            flat = torch.tensor(TR.flatten(w, spec), device="cuda")
            grads = torch.zeros_like(flat)
            opt = TR.AdamOptimizer(eng, flat.numel(), learning_rate=lambda s_: TR.learning_rate_decay(0.002, s_))
            gstep, _ = trn.train_step(opt, flat, grads, feats, labels, global_step=0, seed=1)        # warm-up
            torch.cuda.synchronize()
            t_tr = time.perf_counter()
            n_tr = 3
            for i in range(n_tr):
                gstep, losses = trn.train_step(opt, flat, grads, feats, labels, global_step=gstep, seed=2 + i)
            torch.cuda.synchronize()
            dt_tr = (time.perf_counter() - t_tr) / n_tr
            out["train_step"] = {"workload": "BASELINE.json configs[3]: teacher-forced training step (forward, backward, clip + Adam, "
                                             "weight refresh), B=32, T_in<=140, 800 frames, profile P, fp32",
                                 "ms_per_step": dt_tr * 1e3, "mel_frames_per_s": Bt * St * r / dt_tr, "steps": n_tr,
                                 "loss": float(losses[2])}
            trn.load_params(torch.tensor(TR.flatten(w, spec), device="cuda"))      # back to the bench weights
            trn.close()
            del flat, grads
        except Exception as ex:                                      # never lose the headline line over the extra leg
            out["train_step"] = {"error": repr(ex)}

    # ---- Tacotron v1 (BASELINE.json configs[4]): CBHG encoder + Bahdanau GRU decoder teacher-forced at B = 32, and the
    # mel -> linear CBHG PostNet of train_postnet.py's model, reference hparams sizes (hparams.py:31-60); forward passes ----
    if rank == 0 and world == 1 and not args.no_train:
        try:
            from tacotron2_amd.hparams import V1_PROFILE
            from tacotron2_amd.tacotron_v1 import EngineV1
            from tacotron2_amd.weights_v1 import init_weights_v1
            d1 = V1_PROFILE
            eng1 = EngineV1(d1, init_weights_v1(d1, 1234 + 5))
            rng1 = np.random.default_rng(1234 + 5)
            B1, Ti1, To1 = 32, 140, 800
            lens1v = rng1.integers(60, Ti1 + 1, B1); lens1v[0] = Ti1
            ids1v = rng1.integers(32, 91, (B1, Ti1))
            tgt1v = torch.tensor(rng1.uniform(0, 1, (B1, To1, d1.num_mels)).astype(np.float32), device="cuda")

            def v1_fwd():
                return eng1.decode_teacher(eng1.encode(ids1v, None), lens1v, tgt1v, teacher_forcing=True)[0]

            def timed(fn, reps=3):
                res = fn(); torch.cuda.synchronize()
                t_ = time.perf_counter()
                for _ in range(reps):
                    res = fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t_) / reps, res
            dt_f, mel1v = timed(v1_fwd)
            dt_p, _ = timed(lambda: eng1.postnet(mel1v))
            # TacotronV1PostNetModel.model_fn(TRAIN) (models.py:204-234; train_postnet.py): PostNet forward with batch statistics,
            # L1 loss, backward, clip + Adam, moving statistics -- whole steps, batch_size 32 (hparams.py:63)
            from tacotron2_amd.tacotron_v1 import ModeKeys, TacotronV1PostNetModel
            w1 = init_weights_v1(d1, 1234 + 5)
            pmodel = TacotronV1PostNetModel(d1, eng1, weights=w1)
            spec1v = torch.tensor(rng1.uniform(0, 1, (B1, To1, d1.num_freq)).astype(np.float32), device="cuda")
            lens_o = rng1.integers(To1 // 2, To1 + 1, B1); lens_o[0] = To1
            msk1v = torch.tensor((np.arange(To1)[None, :] < lens_o[:, None]).astype(np.float32), device="cuda")
            dt_pt, res_pt = timed(lambda: pmodel.model_fn({"mel": mel1v}, {"spec": spec1v, "spec_loss_mask": msk1v}, ModeKeys.TRAIN))
            # SingleSpeakerTacotronV1Model.model_fn(TRAIN) (models.py:40-76): the whole step, batch_size 32 (hparams.py:63)
            from tacotron2_amd.tacotron_v1 import SingleSpeakerTacotronV1Model
            S1 = To1 // d1.outputs_per_step
            tl1 = rng1.integers(To1 // 2, To1 + 1, B1); tl1[0] = To1
            sl1 = (tl1 + d1.outputs_per_step - 1) // d1.outputs_per_step
            lab1 = {"mel": tgt1v, "done": (np.arange(S1)[None, :] >= (sl1[:, None] - 1)).astype(np.float32),
                    "spec_loss_mask": (np.arange(To1)[None, :] < tl1[:, None]).astype(np.float32),
                    "binary_loss_mask": (np.arange(S1)[None, :] < sl1[:, None]).astype(np.float32)}
            tmodel = SingleSpeakerTacotronV1Model(d1, weights=w1, seed=3)
            dt_tr1, res_tr1 = timed(lambda: tmodel.model_fn({"source": ids1v, "source_length": lens1v}, lab1, ModeKeys.TRAIN))
            v1_train_loss = float(res_tr1["loss"])
            tmodel.close()
            out["v1"] = {"workload": "BASELINE.json configs[4]: Tacotron v1 (tacotron_v1.py), reference hparams sizes, B=32, T_in<=140, "
                                     "800 frames: EncoderV1 (CBHG) + teacher-forced DecoderV1 forward; mel -> linear CBHG PostNet forward; "
                                     "PostNet training step (forward, L1 loss, backward, clip + Adam, moving statistics); "
                                     "SingleSpeakerTacotronV1Model training step (encoder + decoder forward and backward, clip + Adam)",
                         "teacher_forced_ms": dt_f * 1e3, "teacher_forced_mel_frames_per_s": B1 * To1 / dt_f,
                         "postnet_ms": dt_p * 1e3, "postnet_spectrogram_frames_per_s": B1 * To1 / dt_p,
                         "postnet_train_step_ms": dt_pt * 1e3, "postnet_train_spectrogram_frames_per_s": B1 * To1 / dt_pt,
                         "postnet_train_loss": float(res_pt["loss"]),
                         "train_step_ms": dt_tr1 * 1e3, "train_mel_frames_per_s": B1 * To1 / dt_tr1, "train_loss": v1_train_loss}
            eng1.close()
        except Exception as ex:
            out["v1"] = {"error": repr(ex)}

    # ---- CPU baseline: the NumPy oracle on this box's host cores, bounded sample (N=1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import t2_oracle as O
        try:
            from threadpoolctl import threadpool_info
            cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        except Exception:
            cores = os.cpu_count() or 1
        cores = min(cores, 16)
        sample_steps = S              # the whole step once: ~10-20 s of NumPy work on the box's host cores
        t_a = time.perf_counter()
        mem_o = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
        t_b = time.perf_counter()
        mel_o, _, _ = O.decoder_v2(d, w, mem_o, memory_sequence_length=lens, max_iters=sample_steps)
        t_c = time.perf_counter()
        post_cpu = O.postnet_v2(d, w, mel_o)
        t_d = time.perf_counter()
        scale = S / sample_steps
        cpu_total = (t_b - t_a) + ((t_c - t_b) + (t_d - t_c)) * scale
        cpu_model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except Exception:
            pass
        out["cpu_baseline"] = {"value": B * S * r / cpu_total, "unit": "mel frames/s", "cores": int(cores),
                               "kind": "port", "cpu_model": cpu_model, "os_cpu_count": os.cpu_count(),
                               "sample": "NumPy oracle (CPU restatement, not TensorFlow), the same B=%d batch once: encoder "
                                         "(%.2fs) + %d of %d decoder steps (%.2fs) + PostNetV2 on those %d frames/utt "
                                         "(%.2fs); scale x%.0f" % (B, t_b - t_a, sample_steps, S, t_c - t_b, sample_steps * r, t_d - t_c, scale)}
        # parity spot check of the timed configuration: first sample_steps steps of the GPU decode
        got = bufs[0][:, :sample_steps * r].cpu().numpy()
        memory_g = eng.encode(ids, lens)
        eng.decode_free_launch(memory_g, lens, S, d.min_iters, bufs)
        n_g = eng.decode_free_finish()
        got = bufs[0][:, :sample_steps * r].cpu().numpy()
        post_g = eng.postnet(bufs[0][:, :n_g * r]).cpu().numpy()
        out["parity_check"] = {"max_abs_mel_postnet_diff_B64x%d_frames" % (n_g * r): float(np.abs(post_g - post_cpu).max()),
                               "max_abs_mel_diff_first_20_steps": float(np.abs(got[:, :20 * r] - mel_o[:, :20 * r]).max()),
                               "max_abs_mel_diff_all_%d_free_running_steps" % sample_steps: float(np.abs(got - mel_o).max())}
        # the same check where it can fail: Glorot-initialised attention weights give a uniform softmax
        # that hides the attention score; weights.sharpen_attention gives peaked, moving alignments.  24 free-running steps of
        # the same batch, alignments included.
        from tacotron2_amd.weights import sharpen_attention
        w_s = sharpen_attention(w)
        eng_s = Engine(d, w_s, device=local_rank)
        S_s = 24
        mem_s = O.encoder_v2(d, w_s, O.embedding_lookup(ids, w_s["embedding"]), lens)
        mel_s, stop_s, st_s = O.decoder_v2(d, w_s, mem_s, memory_sequence_length=lens, max_iters=S_s)
        al_s = np.stack(st_s[0].alignment_history, 0)
        g_mel, g_stop, g_al, n_s = eng_s.decode_free(eng_s.encode(ids, lens), lens, max_iters=S_s)
        out["parity_check"]["sharp_attention_%d_steps" % S_s] = {
            "alignment_peak_mean": float(al_s.max(-1).mean()), "steps_equal": bool(n_s == S_s),
            "persistent_kernel": bool(eng_s.get_stat("persist_launches") > 0),
            "max_abs_alignment_diff": float(np.abs(g_al.cpu().numpy() - al_s).max()),
            "max_abs_mel_diff": float(np.abs(g_mel.cpu().numpy() - mel_s).max()),
            "max_abs_stop_diff": float(np.abs(g_stop.cpu().numpy() - stop_s).max())}
        eng_s.close()
        pc, ps = out["parity_check"], out["parity_check"]["sharp_attention_%d_steps" % S_s]
        flat_ok = all(v < 1e-5 for k, v in pc.items() if k.startswith("max_abs"))
        sharp_ok = ps["steps_equal"] and all(v < 2e-4 for k, v in ps.items() if k.startswith("max_abs"))
        pc["asserted"] = {"random_init_all_diffs_below": 1e-5, "sharp_attention_all_diffs_below": 2e-4, "pass": bool(flat_ok and sharp_ok)}
        parity_failed = not (flat_ok and sharp_ok)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if parity_failed:                     # the line above is printed first so that the numbers that failed are on record
        raise SystemExit("bench.py: parity_check failed: %s" % json.dumps(out["parity_check"]))


if __name__ == "__main__":
    main()
