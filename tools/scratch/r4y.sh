mkdir -p gpurun_out/r4y
python -m pytest tests -m gpu -x -q > gpurun_out/r4y/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4y/gputest.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-second-dist --no-extras > gpurun_out/r4y/b.json 2> gpurun_out/r4y/b.err; grep -i "steady" gpurun_out/r4y/b.err
