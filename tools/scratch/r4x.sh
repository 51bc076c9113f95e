mkdir -p gpurun_out/r4x
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras > gpurun_out/r4x/b.json 2> gpurun_out/r4x/b.err
grep -i "steady\|cold" gpurun_out/r4x/b.err
