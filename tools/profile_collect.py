"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.csv, <tag>_fetch_calibration.txt, and profiles/counters.json
(what bench.py prints as roofline.traffic / roofline.valu, with the tag and commit they were measured at)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(root, 'profiles')
line = [l for l in open(os.path.join(src, 'bench.json')).read().splitlines() if l.startswith('{')][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, tag + '_bench.json'), 'w'), indent=1)
B = bench['config']['frames_per_submission']
stats = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)
shutil.copy(stats[0], os.path.join(dst, tag + '_kernel_stats.csv'))
if os.path.exists(os.path.join(src, 'fetch_calibration.txt')):
    shutil.copy(os.path.join(src, 'fetch_calibration.txt'), os.path.join(dst, tag + '_fetch_calibration.txt'))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('pmc_FETCH_SIZE', 'pmc_WRITE_SIZE', 'pmc_SQ', 'pmc_SQ2'):
    for f in glob.glob(os.path.join(src, d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES',
         'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_SMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_INST_CYCLES_SALU', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY')


def steady(v):
    top = max(v)                       # full-batch launches; the one-frame geometry launch is far smaller
    v = [x for x in v if x >= 0.8 * top]
    return sum(v) / len(v)


with open(os.path.join(dst, tag + '_pmc_summary.csv'), 'w') as f:
    f.write('kernel,counter,dispatches,avg_value_per_full_dispatch\n')
    for c in names:
        for k in sorted(agg):
            v = agg[k].get(c)
            if v:
                f.write('%s,%s,%d,%.3f\n' % (k, c, len(v), steady(v)))
fast = [k for k in agg if 'k_fast' in k][0]
# k_fast_tasks goes out as one launch per LDS class (round 4): a BATCH's FAST work = the sum over its launches.  Batches in a pass
# = full-size launches of the largest class.
def per_batch(counter):
    v = agg[fast].get(counter) or []
    if not v:
        return 0.0
    full = [x for x in v if x >= 0.8 * max(v)]
    small = [x for x in v if x < 0.8 * max(v) and x >= 0.004 * max(v)]       # the other classes' launches of full batches
    per_small = (sum(small) / len(full)) if full else 0.0                      # (one-frame launches of the latency legs are far below 0.4 %)
    return sum(full) / len(full) + per_small


fetch, write = per_batch('FETCH_SIZE') * 1024, per_batch('WRITE_SIZE') * 1024
valu, waves = per_batch('SQ_INSTS_VALU'), per_batch('SQ_WAVES')
# SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles) a SIMD spends issuing vector instructions: 4 * ACTIVE / INSTS is
# the measured issue cost of the kernel's own instruction mix; GRBM_GUI_ACTIVE is summed over the 8 XCDs
active = per_batch('SQ_ACTIVE_INST_VALU')
gui = per_batch('GRBM_GUI_ACTIVE') / 8.0
factor, calib = 1.0, 'no calibration run'
cal = os.path.join(src, 'fetch_calibration.txt')
if os.path.exists(cal):
    for l in open(cal):
        if l.startswith('calib_b32'):      # the FAST kernel loads one dword per lane
            ratio = float(l.split('ratio')[1])
            factor = 1.0 / ratio if ratio > 0 else 1.0
            calib = ('FETCH_SIZE*1024 / bytes streamed = %.3f for dword-per-lane loads of a 1 GiB buffer (tools/ubench/fetch_calib, '
                     'same profile round): the counter tallies 128-B requests at 64 B for this width too' % ratio)
commit = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
# the whole pipeline: instructions of ALL kernels per full batch = sum over every dispatch that collected the counter / batches among them
# (a counter may have been collected in more than one pass: each counter is normalised by its own number of full FAST launches)
def nbc(counter):
    v = agg[fast].get(counter) or []
    return max(len([x for x in v if x >= 0.8 * max(v)]), 1) if v else 1


def pipe(counter):
    return sum(sum(agg[k].get(counter, [])) for k in agg) / nbc(counter)


nb = nbc('SQ_INSTS_VALU')
pipe_valu, pipe_active = pipe('SQ_INSTS_VALU'), pipe('SQ_ACTIVE_INST_VALU')
pipe_by_kernel = {k.replace('(anonymous namespace)::', '').replace('orbfe::', '')[:40]: int(sum(agg[k].get('SQ_INSTS_VALU', [])) / max(nb, 1)) for k in sorted(agg)
                  if agg[k].get('SQ_INSTS_VALU')}
# the scalar unit: SALU + SMEM instructions of the dominant kernel and of all kernels of a batch (counters of pmc_SQ and pmc_SQ2 are
# collected in different passes of the same command: a counter present in both is averaged over both)
salu, smem, sca = per_batch('SQ_INSTS_SALU'), per_batch('SQ_INSTS_SMEM'), per_batch('SQ_ACTIVE_INST_SCA')
pipe_salu, pipe_smem = pipe('SQ_INSTS_SALU'), pipe('SQ_INSTS_SMEM')
out = {
    'batch': B,
    'kernel': fast,
    'traffic_bytes_per_launch': int(fetch * factor + write),
    'fetch_size_raw_bytes': int(fetch), 'write_size_raw_bytes': int(write), 'fetch_correction_factor': round(factor, 3),
    'fetch_calibration': calib,
    'valu_insts_per_launch': int(valu), 'valu_insts_per_cell_wave': round(valu / waves, 1), 'waves_per_launch': int(waves),
    'valu_cycles_per_inst': round(4.0 * active / valu, 3) if valu and active else None,
    'valu_busy_frac_under_profiler': round(4.0 * active / 1024.0 / gui, 4) if gui and active else None,
    'pipeline_valu_insts_per_batch': int(pipe_valu), 'pipeline_valu_cycles_per_inst': round(4.0 * pipe_active / pipe_valu, 3) if pipe_valu else None,
    'pipeline_valu_insts_per_batch_by_kernel': pipe_by_kernel,
    'salu_insts_per_launch': int(salu), 'smem_insts_per_launch': int(smem), 'salu_insts_per_cell_wave': round(salu / waves, 1) if waves else None,
    'scalar_active_quadcycles_per_launch': int(sca),
    # tools/ubench/salu_rate.hip (profiles/r05_salu_rate.txt): a SIMD's scalar port takes one SALU instruction per 4.2 - 4.9 cycles whatever
    # the opcode and the number of waves (a CU's one scalar ALU serves its four SIMDs round-robin); 4.25 = the 4- and 8-wave rows
    'salu_cycles_per_inst': 4.25,
    'pipeline_salu_insts_per_batch': int(pipe_salu), 'pipeline_smem_insts_per_batch': int(pipe_smem),
    'source': 'profiles/%s_pmc_summary.csv (rocprofv3 --pmc, separate passes over `python3 bench.py`), measured at commit %s' % (tag, commit),
}
json.dump(out, open(os.path.join(dst, 'counters.json'), 'w'), indent=1)
print(json.dumps(bench)[:300])
print(out)
