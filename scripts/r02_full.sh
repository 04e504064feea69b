cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
( time timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep "^\[bench\]" $O/bench_default.err | cut -c1-260; tail -4 $O/bench_default.err
timeout -k 10 300 python3 bench.py --force-dist --scale 0.05 --workload c3 --steps 3 --warmup 1 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "force-dist rc=$?"; tail -3 $O/bench_forcedist.err | cut -c1-300
