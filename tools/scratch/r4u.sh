set -e
mkdir -p gpurun_out/r4u
python -m pytest tests/test_hip_distributed.py -x -q > gpurun_out/r4u/dist.log 2>&1 || { tail -30 gpurun_out/r4u/dist.log; exit 1; }
tail -3 gpurun_out/r4u/dist.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4u/prof -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-dist --no-extras > $R/gpurun_out/r4u/bench_prof.json 2> $R/gpurun_out/r4u/bench_prof.err
cd $R
f=$(find gpurun_out/r4u/prof -name "*kernel_trace.csv" | head -1)
for k in 0 -3; do python tools/step_timeline.py $f --step $k > gpurun_out/r4u/timeline_$k.txt; done
rm -rf gpurun_out/r4u/prof
head -5 gpurun_out/r4u/timeline_0.txt
