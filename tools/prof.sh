#!/bin/bash
# tools/prof.sh <name> [bench args...]: rocprofv3 kernel stats of bench.py into gpurun_out/<name>, prints the summary
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$name -o p -- python $root/bench.py --no-cpu-baseline --no-batch1 --no-train "$@" > $root/gpurun_out/$name.log 2>&1 || { tail -n 5 $root/gpurun_out/$name.log; exit 1; }
cd $root && python profiles/summarize.py gpurun_out/$name 2>&1 | head -n 14
