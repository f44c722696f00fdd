"""The library build is reproducible (csrc/Makefile: -cuid, -ffile-prefix-map) and its sha256 for the committed sources is recorded in
latent2im_amd/csrc/BUILD_HASHES.json (tools/update_build_hash.py).  This test keeps the record in step with the sources and — when the library
in the tree was built by the recorded compiler — checks that it IS the recorded build."""
import os
import subprocess

import pytest

from latent2im_amd import _lib


def test_build_record_follows_the_kernel_sources():
    rec = _lib.committed_build()
    assert rec is not None, 'latent2im_amd/csrc/BUILD_HASHES.json is missing: python tools/update_build_hash.py'
    assert rec['sources_match'], ('the kernel sources changed (now %s, recorded %s): rebuild and run python tools/update_build_hash.py'
                                  % (_lib.source_hash(), rec['kernel_sources_sha256_16']))


@pytest.mark.skipif(not os.path.isfile(_lib.LIB_PATH), reason='library not built')
def test_the_library_in_the_tree_is_the_recorded_build():
    rec = _lib.committed_build()
    assert rec is not None
    out = subprocess.run(['/opt/rocm/bin/hipcc', '--version'], capture_output=True, text=True).stdout
    here = ' | '.join(l.strip() for l in out.splitlines() if l.startswith(('HIP version', 'AMD clang version')))
    if here != rec['hipcc']:
        pytest.skip('another compiler than the recorded one (%s): the hash is not expected to match' % here)
    assert rec['library_match'], 'libl2i_hip.so in the tree is not what the committed sources build to: rebuild (python __graft_entry__.py build)'
