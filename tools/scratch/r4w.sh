set -e
mkdir -p gpurun_out/r4w
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "new" "base --engine-opt SHARED_FORK=0 --engine-opt AMAX_AHEAD=0"; do
  set -- $cfg; name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4w/prof_$name -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-dist --no-extras "$@" > $R/gpurun_out/r4w/bench_$name.json 2> $R/gpurun_out/r4w/bench_$name.err
  f=$(find $R/gpurun_out/r4w/prof_$name -name "*kernel_trace.csv" | head -1)
  for k in 0 -3; do python3 $R/tools/step_timeline.py $f --step $k > $R/gpurun_out/r4w/timeline_${name}_$k.txt; done
  rm -rf $R/gpurun_out/r4w/prof_$name
done
