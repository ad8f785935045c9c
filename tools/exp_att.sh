#!/bin/bash
# tuning: attention launch time as a function of the Z2 tiles riding in it (option sp_att)
cd "$(dirname "$0")/.."
for n in "$@"; do
  bash tools/prof.sh p_a$n --opt sp_att=$n | grep -E "attention|dense_step_kernel<4, 4>|lstm_step_kernel<4, false, [234]>"
  python - "$n" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/p_a%s.log" % sys.argv[1]) if x.startswith("{")]
d=json.loads(l[-1]); print("sp_att=%s frames/s=%.0f" % (sys.argv[1], d["value"]))
PY
done
