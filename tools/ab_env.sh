#!/bin/bash
# Same-box A/B of tuning-build switches (MI_* environment variables read by tools/probe/libmi355x_rec_tuning.so only):
#   bash tools/ab_env.sh 2 "MI_SORT_FUSED=0" "MI_SORT_FUSED=1" "MI_SORT_FUSED=1 MI_CATCHUP_DEPTH=2"
# alternates the settings ROUNDS times (boxes differ by up to 10 %: only same-box numbers compare) and prints one line each.
ROUNDS=$1; shift
for i in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    env MI_TUNING_LIB=1 $v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras $BENCH_FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d['summary']; k=s['kernel_ms_per_step']
print('%-44s %.3f ms/step  apply %.3f  catchup %.3f (frac %.3f)  gather_frac %.3f (alone %.3f)  gemm %.3f  sort %.3f  bygap %.3f' % ('$v', s['ms_per_step'], s.get('sparse_apply_ms',0), s.get('catchup_ms',0), s.get('catchup_frac',0), s['gather_frac'], s.get('gather_frac_without_side_stream',0), s['gemm_ms_per_step'], k.get('mi_sort_unique_fields/next batch, side stream',0), k.get('mi_catchup_rows_by_gap/next batch, side stream',0)))"
  done
done
