#!/bin/bash
# Same-box A/B of engine.DeepFM scheduling attributes inside the row-sharded one-rank step (one communicator, C = 2):
#   bash tools/shard_opt_ab.sh 2 "" "BYGAP_AHEAD=0" "LIN_SIDE=0"
ROUNDS=$1; shift
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-dist --no-extras --force-shard --chunks 2 --chunk-compute 0 --route-ahead ${ROUTE_AHEAD:-0}"
for i in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    opts=""; for o in $v; do opts="$opts --engine-opt $o"; done
    $B $opts 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%-28s %.3f ms/step' % ('${v:-default}', d['ms_per_step']))"
  done
done
