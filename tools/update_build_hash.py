"""python tools/update_build_hash.py: records, in latent2im_amd/csrc/BUILD_HASHES.json, the sha256 of the libl2i_hip.so that `make` produces from the
committed kernel sources with this image's hipcc.  The build is reproducible (csrc/Makefile: -cuid=<file stem>, -ffile-prefix-map): a reviewer who
builds a clean checkout anywhere gets the same library, and bench.py / tests/test_build_hash_cpu.py compare the library they load with this record —
so "the shipped binary is the committed source" can be checked instead of believed.  Run it after `python __graft_entry__.py build` whenever a file
under csrc/ or include/l2i.h changed (the CPU test fails until the record follows the sources)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latent2im_amd import _lib                            # noqa: E402


def hipcc_version():
    out = subprocess.run(['/opt/rocm/bin/hipcc', '--version'], capture_output=True, text=True).stdout
    return ' | '.join(l.strip() for l in out.splitlines() if l.startswith(('HIP version', 'AMD clang version')))


if __name__ == '__main__':
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'latent2im_amd', 'csrc'), '-j%d' % min(8, os.cpu_count() or 4)], stdout=subprocess.DEVNULL)
    with open(_lib.LIB_PATH, 'rb') as f:
        lib = hashlib.sha256(f.read()).hexdigest()
    rec = dict(kernel_sources_sha256_16=_lib.source_hash(), library_sha256=lib, hipcc=hipcc_version())
    with open(_lib.BUILD_HASHES, 'w') as f:
        json.dump(rec, f, indent=1)
        f.write('\n')
    print(rec)
