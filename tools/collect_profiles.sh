#!/bin/bash
# Run on the GPU box (through gpurun): the default bench line, the same command under rocprofv3 --kernel-trace --stats,
# and two separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the HBM traffic. Output: gpurun_out/prof_<tag>/, which
# tools/update_profiles.py condenses into profiles/.
#   usage: bash tools/collect_profiles.sh r01e
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py 2> $D/bench.err | tail -1 > $D/bench.json
rocprofv3 --kernel-trace --stats -d $D -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu 2> $D/kt.err | tail -1 > $D/bench_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu > $D/fetch.out 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D -o write -- python bench.py --steps 3 --warmup 1 --no-cpu > $D/write.out 2>&1
ls -la $D | head -30
