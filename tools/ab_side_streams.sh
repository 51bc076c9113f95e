#!/bin/bash
# A/B of the side-stream arrangement on one box, alternating (box-to-box differences, 2-4 %, are larger than the effects):
# MI_BYGAP_AHEAD (the next batch's staleness order made ahead) x MI_LIN_SIDE (the wide part's catch-up on its own stream)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
 for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  MI_BYGAP_AHEAD=$1 MI_LIN_SIDE=$2 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.4f ms" % d["ms_per_step"], "gather %.1f us" % (d["roofline"]["avg_launch_ms"]*1e3), "gemm %.3f" % d["roofline_mlp"]["gemm_ms_per_step"], flush=True)
PY
 done
done
