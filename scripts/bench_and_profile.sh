# default bench line + rocprofv3 kernel-trace summary + PMC traffic passes of the same command -> gpurun_out/r01
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; rm -rf $O; mkdir -p $O; cd $R
( time timeout -k 10 900 python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err
grep "^\[bench\]" $O/bench_default.err | cut -c1-200
B="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -- $B > $O/trace_c5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c5 -- $B > $O/pmc_fetch_c5.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write_c5 -- $B > $O/pmc_write_c5.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc_ea_c5 -- $B > $O/pmc_ea_c5.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq_c5 -- $B > $O/pmc_sq_c5.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_lds_c5 -- $B > $O/pmc_lds_c5.log 2>&1
cat $O/trace_c5/*/*_kernel_stats.csv | cut -c1-160 | head -8
