# Round summary run on the GPU box: GPU tests, default bench, kernel trace + PMC passes per workload -> gpurun_out/<round>c;
# condensed afterwards (in the repo) by scripts/make_summary.py <round> into profiles/<round>_summary.md + profiles/traffic.json
#   usage: bash scripts/final_profile.sh r03
cd /tmp && export TMPDIR=/tmp
RND=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${RND}c; rm -rf $O; mkdir -p $O; cd $R
sha256sum spmv_openmp_cuda_amd/lib/libspmvhip.so | cut -d' ' -f1 > $O/libspmvhip.sha256
( time timeout -k 10 900 python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err
grep "^\[bench\]" $O/bench_default.err | cut -c1-200
prof() {  # tag, passes ("t f w s" subset), bench args...
  tag=$1; passes=$2; shift 2
  B="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 $*"
  case "$passes" in *t*) timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/trace -- $B > $O/$tag.trace.log 2>&1;; esac
  case "$passes" in *f*) timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag/pmc_fetch -- $B > $O/$tag.pmc1.log 2>&1;; esac
  case "$passes" in *w*) timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$tag/pmc_write -- $B > $O/$tag.pmc2.log 2>&1;; esac
  case "$passes" in *s*) timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/$tag/pmc_sq -- $B > $O/$tag.pmc3.log 2>&1;; esac
  echo "profiled $tag ($passes)"
}
# the DEFAULT command under the kernel trace (what the driver runs; its per-kernel averages must agree with the line's HIP-event times)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/default/trace -- python3 bench.py > $O/default.trace.json 2> $O/default.trace.log
echo "profiled default"
prof c5 tfws --workload c5
prof c3 tfws --workload c3
prof c3_tiles tfw --workload c3 --launcher hipSpMVTilesCSR
prof c3_onepass tf --workload c3 --launcher hipSpMVWarpPerRowCSR
prof c2_stripes tfw --workload c2 --launcher hipSpMVStripesCSR
prof c2_sell tfw --workload c2 --launcher hipSpMVRowsSELL
prof c3b tfw --workload c3b
# the serial-order default of hipSpMVRowsCSR (deterministic two-phase / stripes forms)
prof c5_serial tfw --workload c5 --launcher hipSpMVRowsCSR --variant 2
prof c3_serial tfw --workload c3 --launcher hipSpMVRowsCSR --variant 2
prof c2_serial tfw --workload c2 --launcher hipSpMVRowsCSR --variant 2
# the reference's kind of matrix: every CSR and ELL kernel on the 3-D stencil stand-in (and the road network)
pstruct() {
  tag=$1; shift
  B="python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --only-structured $*"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/trace -- $B > $O/$tag.trace.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag/pmc_fetch -- $B > $O/$tag.pmc1.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$tag/pmc_write -- $B > $O/$tag.pmc2.log 2>&1
  echo "profiled $tag"
}
pstruct stencil 'stencil3d-500x100x100$'
pstruct road 'road-12M$'
# the same shapes as pattern files (all values 1.0: the kernels stream no values)
pstruct stencil_pattern 'stencil3d-500x100x100-pattern$'
pstruct road_pattern 'road-12M-pattern$'
ls $O
