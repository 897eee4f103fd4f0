# GPU box: k_sfi_resolve's duration in the pipeline (rocprofv3 kernel trace of a short bench run) for LDS candidate pools of 8192 / 0
# entries (ORBFE_SFI_LDS_POOL), with the FAST kernel's generic prologue (ORBFE_FAST_LEAN=0: the configuration whose profile showed
# 500 us outliers, profiles/r05b_kernel_stats.csv) and with the lean one.  A 60 KB block has to wait for a 60 KB hole in a CU's LDS
# while other batches' FAST waves (5 KB each) come and go.
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for lean in 1; do for pool in 8192 0; do for thr in 512 256; do for rep in 1 2 3; do
  export ORBFE_SFI_LDS_POOL=$pool ORBFE_FAST_LEAN=$lean ORBFE_SFI_THREADS=$thr
  d=$root/gpurun_out/sfi_ab/l${lean}_p${pool}_t${thr}_r$rep
  rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $root/bench.py --steps 4 --warmup 1 --passes 2 --cpu-frames 0 --no-pcie --no-latency --prewarm-seconds 0 > /dev/null 2>&1
  python3 - <<PY
import glob,csv
f=glob.glob("$d/**/*kernel_trace.csv",recursive=True)[0]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "sfi_resolve" in r["Kernel_Name"]]
import statistics
print("FAST lean $lean, LDS pool $pool, threads $thr, run $rep: k_sfi_resolve n=%d mean %.1f us max %.1f us sd %.1f  (> 2 x mean: %d)" % (len(d), statistics.mean(d), max(d), statistics.pstdev(d), sum(1 for x in d if x > 2*statistics.mean(d))))
PY
done; done; done; done
