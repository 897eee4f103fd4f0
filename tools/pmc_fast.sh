# SQ / SQC counters of the FAST kernel for one setting of the environment (three --pmc passes over tools/quick_bench.py 32).
# usage: bash tools/pmc_fast.sh <outdir> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
out=${1:-/tmp/pmc_fast}; [ $# -gt 0 ] && shift
for kv in "$@"; do export "$kv"; done
rm -rf $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 /root/repo/tools/quick_bench.py 32 > $out.p$i.log 2>&1 || tail -3 $out.p$i.log
done
python3 /root/repo/tools/pmc_summary.py $out k_fast_tasks
