# rocprofv3: kernel trace + PMC passes for the CSR kernels on c3 / c3b
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof1
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
cd $R
B="python3 bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 2"
for wl in c3 c3b; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- $B --workload $wl > $O/trace_$wl.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc1_$wl -- $B --workload $wl > $O/pmc1_$wl.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc2_$wl -- $B --workload $wl > $O/pmc2_$wl.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3_$wl -- $B --workload $wl > $O/pmc3_$wl.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc4_$wl -- $B --workload $wl > $O/pmc4_$wl.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc5_$wl -- $B --workload $wl > $O/pmc5_$wl.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/pmc6_$wl -- $B --workload $wl > $O/pmc6_$wl.log 2>&1
done
find $O -name "*.csv" | head -50
du -sh $O
