mkdir -p gpurun_out/r4z
python -m pytest tests/test_hip_planes.py -x -q -m gpu > gpurun_out/r4z/planes.log 2>&1; echo "planes rc=$?"; tail -3 gpurun_out/r4z/planes.log
python tools/gemm_pl_timeline.py fwd2 dgrad2 fwd3 dgrad3 > gpurun_out/r4z/tl.txt 2>&1
grep -v "^  [ 0-9][0-9] " gpurun_out/r4z/tl.txt
python tools/mlp_tail_bench.py > gpurun_out/r4z/tail.txt 2>&1; tail -25 gpurun_out/r4z/tail.txt
