"""Guard for the packed-fp32 finding of round 3: on MI355X, v_pk_{fma,mul,add}_f32 with op_sel set on SRC1 (src1's high half
feeding the low result) returns sporadically wrong results while a bf16-MFMA kernel is resident on the same CUs.  hipcc's SLP vectorizer emits that
form; the library is built with -fno-slp-vectorize and its hand-written packed code keeps half selections on src0 / in op_sel_hi.  This test
disassembles the gfx950 code objects inside the built libl2i_hip.so and fails if the form is back."""
import os
import re
import subprocess
import tempfile

import pytest

from latent2im_amd import _lib

LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


@pytest.mark.skipif(not os.path.isfile(_lib.LIB_PATH) or not os.path.isfile(os.path.join(LLVM, 'llvm-objdump')), reason='library or LLVM tools absent')
def test_no_packed_fp32_with_op_sel_on_src1():
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fatbin')
        # an explicit output file: without one llvm-objcopy rewrites its INPUT in place (same code, different bytes: the library hash that bench.py
        # matches against profiles/*_hbm_traffic.json would change under the test)
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, _lib.LIB_PATH, os.path.join(td, 'discard.so')])
        blob = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert len(starts) >= 10, 'expected one offload bundle per object file, found %d' % len(starts)
        bad, packed, kernels = [], 0, 0
        for i, s in enumerate(starts):
            piece = os.path.join(td, 'b%d' % i)
            open(piece, 'wb').write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, 'co%d' % i)
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                                   '--input=' + piece, '--output=' + co], stderr=subprocess.DEVNULL)
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], capture_output=True, text=True).stdout
            cur = '?'
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
                if m:
                    cur = m.group(1)
                    kernels += 1
                    continue
                if re.search(r'\bv_pk_(fma|mul|add)_f32\b', line):
                    packed += 1
                    sel = re.search(r'op_sel:\[([01]),([01])', line)
                    if sel and sel.group(2) == '1':                         # op_sel[1] = src1's high half for the LOW result
                        bad.append((cur[:60], line.strip()[:100]))
        assert kernels > 100 and packed > 0, (kernels, packed)              # the disassembly really covered the library (its Winograd / h8 epilogues use packed code)
        assert not bad, 'packed fp32 with op_sel on src1 (unsafe beside bf16 MFMA on gfx950): %d instructions, e.g. %s' % (len(bad), bad[:3])
