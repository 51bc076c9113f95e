mkdir -p gpurun_out/r5c
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d['summary']; k=d['kernel_ms_per_step']
print('%-40s %.3f ms/step  apply %.3f  catchup %.3f  gather_frac %.3f  gemm %.3f  dgrad %.3f  fwd %.3f wgrad %.3f' % ('$*', s['ms_per_step'], s.get('sparse_apply_ms',0), s.get('catchup_ms',0), s['gather_frac'], s['gemm_ms_per_step'], k['mi_dense_bwd_data_planes'], k['mi_dense_fwd_planes'], k['mi_dense_bwd_weight_planes_batch']))"; }
for i in 1 2 3 4; do
  run --engine-opt BYGAP_WHERE=0
  run --engine-opt BYGAP_WHERE=1
  run --engine-opt BYGAP_WHERE=2
done 2>&1 | tee gpurun_out/r5c/ab.txt
