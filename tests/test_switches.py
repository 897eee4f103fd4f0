"""Every environment variable the shipped library reads is either plain configuration or a route that a `-m gpu` parity test
runs against the oracle; measurement switches exist only in a `make EXPERIMENTS=1` build."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = sorted(glob.glob(os.path.join(ROOT, 'os1_amd', 'csrc', '*')) + glob.glob(os.path.join(ROOT, 'include', '*.h')) +
                 glob.glob(os.path.join(ROOT, 'include', 'orbfe', '*')))
SOURCES = [p for p in SOURCES if p.endswith(('.hip', '.cpp', '.h', '.hpp', '.inc'))]

# configuration a deployment sets, not a code route: which GPU, how many host threads, whether to print, what the HIP runtime was given
CONFIGURATION = {'GPU_MAX_HW_QUEUES', 'ORBFE_DEVICE', 'ORBFE_QUIET', 'ORBFE_HOST_THREADS'}
MAX_SWITCHES = 25


def _names(pattern):
    out = {}
    for p in SOURCES:
        for m in re.finditer(pattern, open(p, errors='replace').read()):
            out.setdefault(m.group(1), set()).add(os.path.relpath(p, ROOT))
    return out


def _gpu_test_text():
    text = ''
    for p in sorted(glob.glob(os.path.join(ROOT, 'tests', 'test_gpu_*.py')) + glob.glob(os.path.join(ROOT, 'tests', 'test_natural_images.py'))):
        text += open(p).read()
    return text


def test_every_runtime_switch_is_configuration_or_has_a_gpu_parity_test():
    product = _names(r'getenv\("([A-Z_0-9]+)"\)')
    assert len(product) <= MAX_SWITCHES, sorted(product)
    tests = _gpu_test_text()
    untested = [n for n in sorted(product) if n not in CONFIGURATION and ("'%s'" % n) not in tests and ('"%s"' % n) not in tests]
    assert not untested, 'runtime switches without a -m gpu test: %s' % untested
    # the list DESIGN.md s4 prints is this list
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    missing = [n for n in sorted(product) if n not in design]
    assert not missing, 'switches DESIGN.md does not mention: %s' % missing


def test_measurement_switches_exist_only_in_the_experiments_build():
    product = _names(r'getenv\("([A-Z_0-9]+)"\)')
    lab = _names(r'ORBFE_EXP_ENV\("([A-Z_0-9]+)"\)')
    assert lab and not (set(lab) & set(product)), sorted(set(lab) & set(product))
    # no source file reads the environment any other way
    for p in SOURCES:
        src = open(p, errors='replace').read()
        assert 'secure_getenv' not in src and not re.search(r'\benviron\b', src), p
        for m in re.finditer(r'getenv\(([^)]*)\)', src):
            arg = m.group(1).strip()
            assert re.fullmatch(r'"[A-Z_0-9]+"', arg) or arg in ('name',), (p, m.group(0))   # (the macro's own definition passes `name`)
    so = os.path.join(ROOT, 'os1_amd', 'liborbfe.so')
    if os.path.exists(so):   # the default build does not even carry their names
        strings = subprocess.run(['strings', '-n', '8', so], capture_output=True, text=True).stdout
        for n in sorted(lab):
            assert n not in strings, '%s is compiled into the default liborbfe.so' % n
        for n in sorted(product):
            if not all(f.startswith('include/orbfe/') for f in product[n]):   # (header-only shim switches are compiled into the caller)
                assert n in strings, n
