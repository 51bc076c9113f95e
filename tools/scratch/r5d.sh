mkdir -p gpurun_out/r5d
python -m pytest tests/test_hip_kernels.py tests/test_abi.py -x -q -m gpu -k "gap or abi or catchup" > gpurun_out/r5d/t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r5d/t.log
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d['summary']; k=d['kernel_ms_per_step']
print('%-40s %.3f ms/step  apply %.3f  catchup %.3f  gather_frac %.3f  gemm %.3f  dgrad %.3f  exact %.3f' % ('$*', s['ms_per_step'], s.get('sparse_apply_ms',0), s.get('catchup_ms',0), s['gather_frac'], s['gemm_ms_per_step'], k['mi_dense_bwd_data_planes'], d.get('catchup_exact',{}).get('ms_per_step',0)))"; }
for i in 1 2 3 4; do
  run --engine-opt GAP_GROUPS=1
  run --engine-opt GAP_GROUPS=0
done 2>&1 | tee gpurun_out/r5d/ab.txt
