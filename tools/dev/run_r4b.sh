set -x
mkdir -p gpurun_out/r4b
timeout 1200 python -m pytest tests/test_gpu_bench.py tests/test_gpu_hostbatch.py -x -q > gpurun_out/r4b/tests.txt 2>&1
tail -5 gpurun_out/r4b/tests.txt
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err
tail -c 3000 gpurun_out/r4b/bench.json; tail -5 gpurun_out/r4b/bench.err
