"""tools/collect_profiles.py <tag> <round> [commit]: copy the evidence `tools/evidence.sh <tag>` left under gpurun_out/ into
profiles/<round>_* with a header naming the commit and the build it was captured on, and record the persistent decoder's
memory-side traffic in profiles/traffic.json (the figure bench.py's `roofline.traffic` quotes).  Runs here, after the gpurun
call (the GPU box has no .git)."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag, rnd = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "tacotron2_amd/csrc", "bench.py"]).decode().strip()
so = hashlib.sha256(open(os.path.join(ROOT, "tacotron2_amd", "libt2mi.so"), "rb").read()).hexdigest()[:16]
head = (f"# captured on commit {commit}{' + uncommitted changes to csrc/ or bench.py' if dirty else ''}; libt2mi.so sha256 {so}; "
        f"tools/evidence.sh {tag} on one MI355X (gpurun)\n")
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def put(src, dst, comment=True):
    body = open(os.path.join(G, src)).read()
    with open(os.path.join(P, dst), "w") as f:
        f.write((head if comment else "") + body)
    print("wrote profiles/" + dst)


put(f"{tag}_bench_kernel_stats.txt", f"{rnd}_bench_kernel_stats.txt")
put(f"{tag}_pmc.txt", f"{rnd}_pmc.txt")
if os.path.exists(os.path.join(G, f"{tag}_v1train_kernel_stats.txt")):
    put(f"{tag}_v1train_kernel_stats.txt", f"{rnd}_v1_train_kernel_stats.txt")
line = [l for l in open(os.path.join(G, f"{tag}_bench_line_under_rocprof.json")) if l.startswith("{")][-1]
d = json.loads(line)
d["_captured"] = head.strip("# \n")
json.dump(d, open(os.path.join(P, f"{rnd}_bench_line_under_rocprof.json"), "w"), indent=1)
print("wrote profiles/%s_bench_line_under_rocprof.json" % rnd)

# ---- traffic.json ----
pmc = open(os.path.join(G, f"{tag}_pmc.txt")).read()
sections = re.split(r"^== ", pmc, flags=re.M)[1:]


def avg(cmd_part, counter, kernel_part):
    for s in sections:
        first, rest = s.split("\n", 1)
        if cmd_part in first and counter in first.split(" -- ")[0]:
            for l in rest.splitlines():
                if kernel_part in l and f"'{counter}'" in l:
                    return float(l.rsplit("avg=", 1)[1])
    raise SystemExit(f"no {counter} of {kernel_part} in the '{cmd_part}' pass")


tj = json.load(open(os.path.join(P, "traffic.json")))
tj["source"] = "one entry per round and kernel below; each names its own passes.  Top-level r01 keys: round 1's launch-per-phase LSTM kernels (profiles/r01_pmc_fetch_write.txt)"


def entry(cmd, kernel, steps, alg_per_step, note):
    f, w = avg(cmd, "FETCH_SIZE", kernel), avg(cmd, "WRITE_SIZE", kernel)
    fb, wb = 2.0 * f * 1000.0, w * 1000.0        # KB per dispatch; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)
    e = {"source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python {cmd}`; profiles/{rnd}_pmc.txt; {head.strip('# ').strip()}",
         "kernel": kernel, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
         "hbm_bytes_per_step": (fb + wb) / steps, "note": note}
    if alg_per_step:
        e["algorithmic_bytes_per_step"] = alg_per_step
        e["ratio"] = (fb + wb) / steps / alg_per_step
    return e


old64, old1 = tj.get("r02_persistent_decode_b64", {}), tj.get("r02_persistent_decode_b1", {})
tj[f"{rnd}_persistent_decode_b64"] = entry("tools/b1_run.py B=64", "pz::decode_kernel<4, 0>", 400, old64.get("algorithmic_bytes_per_step"),
                                           "one launch = 400 decoder steps at B = 64, T_in = 100..140 (bench seed); memory-side counters, Infinity-Cache hits included")
tj[f"{rnd}_persistent_decode_b1"] = entry("tools/b1_run.py B=1", "pz::decode_kernel<1, 0>", 400, old1.get("algorithmic_bytes_per_step"),
                                          "one launch = 400 decoder steps at B = 1, T_in = 100")
tj[f"{rnd}_train_forward_b32"] = entry("tools/bwd_stamps.py B=32", "pz::decode_kernel<2, 4>", 400, None, "teacher-forced training forward (two-chain schedule), 400 steps at B = 32 (tape written)")
tj[f"{rnd}_train_backward_b32"] = entry("tools/bwd_stamps.py B=32", "pz::backward_kernel<2>", 400, None, "reverse-time recurrence, 400 steps at B = 32")
tj["current"] = f"{rnd}_persistent_decode_b64"
json.dump(tj, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print("updated profiles/traffic.json:", json.dumps({k: round(v["hbm_bytes_per_step"] / 1e6, 1) for k, v in tj.items() if isinstance(v, dict) and k.startswith(rnd)}))
