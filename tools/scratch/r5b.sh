mkdir -p gpurun_out/r5b
python -m pytest tests/test_hip_planes.py -x -q -m gpu > gpurun_out/r5b/planes.log 2>&1; echo "planes rc=$?"; tail -3 gpurun_out/r5b/planes.log
MI_TUNING_LIB=1 python tools/mlp_tail_bench.py > gpurun_out/r5b/tail.txt 2>&1; grep -A20 "plan variants" gpurun_out/r5b/tail.txt
