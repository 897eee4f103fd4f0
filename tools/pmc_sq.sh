# SQ counters of the extractor kernels (two passes); summarise with tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp
out=${1:-/root/repo/gpurun_out/pmc2}
rm -rf $out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$n -- python3 /root/repo/tools/quick_bench.py 32 > /dev/null 2>&1
done
python3 /root/repo/tools/pmc_summary.py $out ${@:2}
