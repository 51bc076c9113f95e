mkdir -p gpurun_out/r4v
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d['summary']
print('%-60s %.3f ms/step  apply %.3f  catchup %.3f  gather_frac %.3f  gemm %.3f' % ('$*', s['ms_per_step'], s.get('sparse_apply_ms',0), s.get('catchup_ms',0), s['gather_frac'], s['gemm_ms_per_step']))"; }
for i in 1 2 3; do
  run --engine-opt SHARED_FORK=0 --engine-opt AMAX_AHEAD=0
  run
  run --engine-opt LIN_FWD_FORK=0
  run --engine-opt SHARED_FORK=0
  run --engine-opt AMAX_AHEAD=0
done 2>&1 | tee gpurun_out/r4v/ab.txt
