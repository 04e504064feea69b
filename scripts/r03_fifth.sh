#!/bin/bash
set -o pipefail
O=gpurun_out/r03_fifth
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ell or fuzz or degenerate or config4" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 600 python bench.py --only-structured all --no-cpu-baseline > $O/structured.json 2> $O/structured.err
echo "bench rc=$?"
grep "ELL" $O/structured.err
