mkdir -p gpurun_out/r5g
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d['summary']; k=d['kernel_ms_per_step']
print('%-30s %.3f ms/step  apply %.3f  catchup %.3f  gather_frac %.3f (alone %.3f)  gemm %.3f  dgrad %.3f fwd %.3f' % ('$*', s['ms_per_step'], s.get('sparse_apply_ms',0), s.get('catchup_ms',0), s['gather_frac'], d['roofline'].get('frac_without_side_stream',0), s['gemm_ms_per_step'], k['mi_dense_bwd_data_planes'], k['mi_dense_fwd_planes']))"; }
for i in 1 2 3 4 5; do
  run --engine-opt GAP_GROUPS=0
  run --engine-opt GAP_GROUPS=1
done 2>&1 | tee gpurun_out/r5g/ab.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5g/prof -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-dist --no-extras --engine-opt GAP_GROUPS=1 > $R/gpurun_out/r5g/bench_g1.json 2> $R/gpurun_out/r5g/bench_g1.err
f=$(find $R/gpurun_out/r5g/prof -name "*kernel_trace.csv" | head -1)
python3 $R/tools/step_timeline.py $f --step -3 > $R/gpurun_out/r5g/timeline_g1.txt
rm -rf $R/gpurun_out/r5g/prof
