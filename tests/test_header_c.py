"""include/orbfe.h is a plain-C header: it must compile as C89-compatible C and as C++ without any other include."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_compiles_as_c_and_cpp():
    with tempfile.TemporaryDirectory() as d:
        for name, lang, std in (('t.c', 'c', '-std=c99'), ('t.cpp', 'c++', '-std=c++11')):
            p = os.path.join(d, name)
            open(p, 'w').write('#include "orbfe.h"\nint main(void) { return (int)sizeof(OrbfeKeyPoint) - 28; }\n')
            subprocess.check_call(['gcc', '-x', lang, std, '-Wall', '-Werror', '-pedantic', '-fsyntax-only',
                                   '-I' + os.path.join(ROOT, 'include'), p])
