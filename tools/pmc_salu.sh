# Scalar-unit counters of the extractor kernels (a pass of their own); summarise with tools/pmc_summary.py.
# usage (GPU box): bash tools/pmc_salu.sh [outdir]      -- blocking 32-frame 1080p batches (tools/quick_bench.py 32)
cd /tmp && export TMPDIR=/tmp
out=${1:-/root/repo/gpurun_out/pmc_salu}
rm -rf $out
rocprofv3 --list-avail 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*(SALU|SCA|SMEM|ISSUE|INST_CYCLES)[A-Z_0-9]*" | sort -u > $out.avail.txt 2>/dev/null
for set in "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$n -- python3 /root/repo/tools/quick_bench.py 32 > $out.$n.log 2>&1
done
python3 /root/repo/tools/pmc_summary.py $out ${@:2}
