mkdir -p gpurun_out/r5e
python tools/catchup_bench.py > gpurun_out/r5e/cb.txt 2>&1; cat gpurun_out/r5e/cb.txt | tail -12
