"""FETCH_SIZE (KB) per calibration kernel of tools/ubench/fetch_calib vs the bytes it reads from HBM."""
import csv, glob, sys
want = {'calib_b128': 1 << 30, 'calib_b32': 1 << 30, 'calib_tiles': (1 << 30) // (1920 * 1080) * 1884 * 1060}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k in want and r['Counter_Name'] == 'FETCH_SIZE':
            v = float(r['Counter_Value']) * 1024
            print('%-12s FETCH_SIZE*1024 = %12.0f   distinct bytes read = %12d   ratio %.3f' % (k, v, want[k], v / want[k]))
