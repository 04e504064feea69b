# Round summary run: tests, default bench, kernel trace + PMC passes (c5 auto, c3 auto, c3n one-pass), config-4 tables
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01c; rm -rf $O; mkdir -p $O; cd $R
( time timeout -k 10 900 python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err
grep "^\[bench\]" $O/bench_default.err | cut -c1-200
prof() {  # tag, bench args...
  tag=$1; shift
  B="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/trace -- $B > $O/$tag.trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag/pmc_fetch -- $B > $O/$tag.pmc1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$tag/pmc_write -- $B > $O/$tag.pmc2.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/$tag/pmc_sq -- $B > $O/$tag.pmc3.log 2>&1
}
prof c5 --workload c5
prof c3 --workload c3
prof c3_onepass --workload c3 --launcher hipSpMVWarpPerRowCSR
prof c3n --workload c3n --launcher hipSpMVWarpPerRowCSR
prof c2 --workload c2 --launcher hipSpMVRowsSELL
timeout -k 10 600 python3 scripts/config4_ell.py > $O/config4_ell.md 2>/dev/null
timeout -k 10 600 python3 scripts/config4_ell.py 1.0 512 > $O/config4_ell_band512.md 2>/dev/null
ls $O
