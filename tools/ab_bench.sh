#!/bin/bash
# A/B of one engine scheduling / layout attribute on ONE box: alternating bench runs (the boxes of the pool differ by up to
# 10 % in HBM speed, so only same-box pairs mean anything).  usage: bash tools/ab_bench.sh NAME [rounds] [extra bench flags]
#   e.g. bash tools/ab_bench.sh ROW_RECORDS 3
NAME=$1; ROUNDS=${2:-3}; shift; shift
for i in $(seq 1 $ROUNDS); do
  for v in 1 0; do
    python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras --engine-opt $NAME=$v "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d['summary']
print('$NAME=$v  %.3f ms/step  apply %.3f  catchup %.3f  gather_frac %.3f  gemm %.3f' % (s['ms_per_step'], s.get('sparse_apply_ms',0), s.get('catchup_ms',0), s['gather_frac'], s['gemm_ms_per_step']))"
  done
done
