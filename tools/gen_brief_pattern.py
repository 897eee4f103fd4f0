#!/usr/bin/env python3
"""Emit the 256x4 rBRIEF sampling table (ORB's learned bit_pattern_31_) as a compact int8 .inc file.

The table is DATA (the ORB paper's learned test locations, identical in OpenCV's orb.cpp and
in the reference at src/ORBextractor.cc:182-440).  This script parses the integers out of the
reference file and writes them, 16 per line, into the two places that need the table:
  oracle/brief_pattern.inc        (CPU oracle)
  os1_amd/csrc/brief_pattern.inc  (HIP product)
A sha256 of the 1024 int8 values is printed and pinned in tests/test_oracle_kat.py.
Run only in the build container (needs /root/reference); the .inc files are committed.
"""
import hashlib, re, sys
import numpy as np

src = open('/root/reference/src/ORBextractor.cc', encoding='utf-8', errors='replace').read()
m = re.search(r'bit_pattern_31_\[256\*4\]\s*=\s*\{(.*?)\};', src, re.S)
body = re.sub(r'/\*.*?\*/', '', m.group(1), flags=re.S)
vals = [int(v) for v in re.findall(r'-?\d+', body)]
assert len(vals) == 1024, len(vals)
a = np.array(vals, dtype=np.int8)
assert a.min() >= -13 and a.max() <= 13
print('sha256', hashlib.sha256(a.tobytes()).hexdigest())
lines = []
for i in range(0, 1024, 16):
    lines.append(','.join('%3d' % v for v in vals[i:i + 16]) + ',')
txt = ("// rBRIEF test locations, 256 x (x1,y1,x2,y2), int8 (data table; see tools/gen_brief_pattern.py)\n"
       + '\n'.join(lines) + '\n')
for out in ('oracle/brief_pattern.inc', 'os1_amd/csrc/brief_pattern.inc'):
    open(out, 'w').write(txt)
