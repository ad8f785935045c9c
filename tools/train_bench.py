"""BASELINE.json configs[3]: Tacotron2 teacher-forced training step, B = 32 per GPU, profile P, T_in <= 140, T_out = 800
(S = 400 decoder steps), random-init weights, synthetic data.  Prints the step time and its parts.
    python tools/train_bench.py [steps]          (single GPU)
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py   (RCCL gradient all-reduce)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import torch.distributed as dist
from tacotron2_amd import training as TR
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights

rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = PROFILE_P
w = init_weights(d, seed=1234 + 4)
eng = Engine(d, w, device=local)
B, T, S = 32, 140, 400
rng = np.random.default_rng(1234 + 4 + 100 * rank)
lens = rng.integers(60, T + 1, B).astype(np.int64); lens[0] = T
ids = np.zeros((B, T), np.int64)
for b, n in enumerate(lens): ids[b, :n] = rng.integers(32, 91, n)
tl = rng.integers(400, 801, B); tl[0] = 800
sl = (tl + 1) // 2
labels = {"mel": rng.uniform(0, 1, (B, S * 2, d.num_mels)).astype(np.float32),
          "done": (np.arange(S)[None, :] >= (sl[:, None] - 1)).astype(np.float32),
          "spec_loss_mask": (np.arange(S * 2)[None, :] < tl[:, None]).astype(np.float32),
          "binary_loss_mask": (np.arange(S)[None, :] < sl[:, None]).astype(np.float32)}
labels = {k: torch.tensor(v, device="cuda") for k, v in labels.items()}
feats = {"source": ids, "source_length": lens}
spec = TR.flatten_spec(w)
tr = TR.Trainer(eng, B, T, S)
tr.bind_gradients(spec)
flat = torch.tensor(TR.flatten(w, spec), device="cuda")
grads = torch.zeros_like(flat)
opt = TR.AdamOptimizer(eng, flat.numel(), learning_rate=lambda s: TR.learning_rate_decay(0.002, s))

def sync():
    torch.cuda.synchronize()
    return time.perf_counter()

step = 0
step, losses = tr.train_step(opt, flat, grads, feats, labels, global_step=step, seed=1)       # warm-up (allocations, packing)
t0 = sync()
for i in range(steps):
    step, losses = tr.train_step(opt, flat, grads, feats, labels, global_step=step, seed=2 + i)
t1 = sync()
dt = (t1 - t0) / steps
# parts (one more step, synchronised between phases)
parts = {}
a = sync(); grads.zero_(); mem, _ = tr.encoder_forward(ids, lens, 0.5, 99); b = sync(); parts["encoder fwd"] = b - a
tr.decoder_forward(mem, lens, labels["mel"], 0.5, 99); c = sync(); parts["decoder fwd"] = c - b
dmem, _ = tr.decoder_backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], grads); e = sync(); parts["decoder bwd"] = e - c
tr.encoder_backward(dmem, grads); f = sync(); parts["encoder bwd"] = f - e
scale = TR.allreduce_gradients(grads); g = sync(); parts["all-reduce"] = g - f
opt.apply_gradients(grads, flat, global_step=step, grad_scale=scale); h = sync(); parts["clip+adam"] = h - g
tr.update_moving_stats(flat); tr.load_params(flat); k = sync(); parts["refresh"] = k - h
if rank == 0:
    print("training step B=%d/GPU x %d GPU(s), T_in=%d, S=%d (800 frames): %.1f ms/step, %.0f mel frames/s (all ranks), loss %.4f"
          % (B, world, T, S, dt * 1e3, world * B * S * 2 / dt, float(losses[2])))
    print("  " + ", ".join("%s %.1f ms" % (k2, v * 1e3) for k2, v in parts.items()))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
