#!/bin/bash
# tools/evidence.sh <tag>: every profile the round's bench line is judged against, on the CURRENT build, into gpurun_out/<tag>_*:
#   1. rocprofv3 --kernel-trace --stats of the whole `python bench.py --no-cpu-baseline` command (+ its JSON line);
#   2. separate --pmc passes (no trace domains): FETCH_SIZE / WRITE_SIZE of the free-running decode at B = 64 and B = 1, and of the
#      training recurrences; GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES of the B = 64 decode and of PostNetV2
#      (effective clock and matrix-pipe occupancy);
#   3. rocprofv3 --kernel-trace --stats of SingleSpeakerTacotronV1Model's training step (tools/v1_train_bench.py).
# Every step has its own time limit; a failing step ends the script.
set -uo pipefail
tag=$1
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
step() { echo "== $*"; }
step "kernel trace of bench.py"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_bench -o p -- python $root/bench.py --no-cpu-baseline > $out/${tag}_bench_line_under_rocprof.json 2> $out/${tag}_bench.err || { tail -n 5 $out/${tag}_bench.err; exit 1; }
python $root/profiles/summarize.py $out/${tag}_bench 40 > $out/${tag}_bench_kernel_stats.txt || exit 1
pmc() {   # name counters... -- script args
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  step "pmc $name: ${ctrs[*]} -- $*"
  timeout -s KILL 240 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $out/${tag}_$name -o p -- python "$@" > $out/${tag}_$name.log 2>&1 || { tail -n 5 $out/${tag}_$name.log; return 1; }
  { echo "== rocprofv3 --pmc ${ctrs[*]} -- python ${*/$root\//}"; python $root/tools/pmc_sum.py $out/${tag}_$name | grep -E "pz::|conv_gemm"; } >> $out/${tag}_pmc.txt
}
: > $out/${tag}_pmc.txt
pmc f64 FETCH_SIZE -- $root/tools/b1_run.py B=64 && \
pmc w64 WRITE_SIZE -- $root/tools/b1_run.py B=64 && \
pmc f1 FETCH_SIZE -- $root/tools/b1_run.py B=1 && \
pmc w1 WRITE_SIZE -- $root/tools/b1_run.py B=1 && \
pmc ftr FETCH_SIZE -- $root/tools/bwd_stamps.py B=32 && \
pmc wtr WRITE_SIZE -- $root/tools/bwd_stamps.py B=32 && \
pmc clk64 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- $root/tools/b1_run.py B=64 && \
pmc clkpost GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- $root/tools/postnet_run.py || exit 1
cat $out/${tag}_pmc.txt
step "kernel trace of the V1 model's training step (tools/v1_train_bench.py)"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_v1train -o p -- python $root/tools/v1_train_bench.py 4 > $out/${tag}_v1train.log 2>&1 || { tail -n 5 $out/${tag}_v1train.log; exit 1; }
{ grep "V1 training step" $out/${tag}_v1train.log; python $root/profiles/summarize.py $out/${tag}_v1train 24; } > $out/${tag}_v1train_kernel_stats.txt || exit 1
