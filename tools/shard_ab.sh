#!/bin/bash
# The row-sharded step's own cost on ONE GPU (a one-rank RCCL group: no link time) beside the single-GPU step, same box:
#   bash tools/shard_ab.sh [rounds]
ROUNDS=${1:-1}
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-dist --no-extras"
for i in $(seq 1 $ROUNDS); do
  for f in "" "--force-shard --chunks 1 --chunk-compute 0" "--force-shard --chunks 2 --chunk-compute 0" "--force-shard --chunks 2 --chunk-compute 0 --route-ahead 0" "--force-shard --chunks 2 --chunk-compute 1"; do
    $B $f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernel_ms_per_step']
top=sorted(k.items(), key=lambda kv:-kv[1])[:12]
print('%-62s %.3f ms/step   %s' % ('${f:-single GPU}', d['ms_per_step'], ' '.join('%s=%.0f' % (a.replace('mi_',''), b*1e3) for a,b in top)))"
  done
done
