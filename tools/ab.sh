#!/bin/bash
# A/B helper for GPU runs: tools/ab.sh "<opts A>" "<opts B>" ...  (each: comma-separated name=value, or "-")
cd "$(dirname "$0")/.."
for o in "$@"; do
  args=""; if [ "$o" != "-" ]; then for kv in ${o//,/ }; do args="$args --opt $kv"; done; fi
  for rep in 1 2; do
    timeout -k 10 120 python bench.py --no-cpu-baseline $args > gpurun_out/ab_tmp.log 2>&1 || { echo "bench failed for $o"; tail -n 5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$o" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/ab_tmp.log") if x.startswith("{")]
d=json.loads(l[-1]); print("%-28s frames/s=%9.0f  lstm_avg_us=%6.2f  b1_ms=%6.2f" % (sys.argv[1], d["value"], d["roofline"]["avg_launch_us"], d.get("batch1",{}).get("ms",0)))
PY
  done
done
