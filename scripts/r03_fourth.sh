#!/bin/bash
# round 3, fourth GPU call: bench with the structured block (small scale first, then full)
set -o pipefail
O=gpurun_out/r03_fourth
mkdir -p $O
timeout -k 10 600 python bench.py --scale 0.02 --steps 5 --warmup 1 > $O/bench_small.json 2> $O/bench_small.err
echo "small bench rc=$?"
tail -5 $O/bench_small.err
timeout -k 10 1000 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
grep -v "^\[bench\] c4\|amdgpu.ids" $O/bench.err | tail -70
