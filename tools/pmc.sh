#!/bin/bash
# tools/pmc.sh <name> <counter> <python script + args...>: one rocprofv3 --pmc pass (own run, no trace domains)
name=$1; ctr=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/$name -o p -- python "$@" > $root/gpurun_out/$name.log 2>&1 || { tail -n 5 $root/gpurun_out/$name.log; exit 1; }
cd $root && python tools/pmc_sum.py gpurun_out/$name | grep -E "decode_kernel|lstm_step|attention_step|dense_step|conv_gemm" | head -n 12
