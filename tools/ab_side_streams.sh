#!/bin/bash
# A/B of the side-stream arrangement on one box, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
 for cfg in "0 0 2048" "1 0 2048" "0 1 2048" "1 1 2048" "1 1 1024" "0 0 1024"; do
  set -- $cfg
  MI_BYGAP_AHEAD=$1 MI_LIN_SIDE=$2 MI_CATCHUP_BLOCKS=$3 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1]); print(sys.argv[1], "%.4f"%d["ms_per_step"], flush=True)
PY
 done
done
