mkdir -p gpurun_out/r5a
python -m pytest tests/test_hip_planes.py tests/test_hip_model.py tests/test_canned_parity.py -x -q -m gpu > gpurun_out/r5a/t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r5a/t.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras > gpurun_out/r5a/b.json 2> gpurun_out/r5a/b.err; grep -i "steady" gpurun_out/r5a/b.err
python - <<'P'
import json
d=[json.loads(l) for l in open('gpurun_out/r5a/b.json') if l.startswith('{')][0]
k=d['kernel_ms_per_step']
for a,b in sorted(k.items(), key=lambda x:-x[1]): print("%-60s %.4f"%(a,b))
print(d['roofline_mlp']['gemm_ms_per_step'], d['roofline_mlp']['frac'])
P
