"""python tools/isa_blocks.py file.s mangled_kernel_substring: per basic block of a kernel's gfx950 assembly, the instruction mix
(MFMA / packed VALU / other VALU / spill lanes / LDS / buffer / scalar) — a quick look at what sits in a hot loop."""
import re
import sys
from collections import Counter


def main(path, key):
    s = open(path).read()
    m = re.search(r'^(\S*%s\S*):' % re.escape(key), s, re.M)
    i = m.start()
    j = s.index('.Lfunc_end', i)
    blocks, cur = [], None
    for l in s[i:j].splitlines():
        if re.match(r'^\.LBB\d+_\d+:', l) or cur is None:
            cur = [l.strip(), []]
            blocks.append(cur)
        else:
            cur[1].append(l.strip())
    for name, ins in blocks:
        ops = [x.split()[0] for x in ins if x and not x.startswith(('.', ';', '/'))]
        if not any(o.startswith('v_mfma') for o in ops) and len(ops) < 40:
            continue
        c = Counter()
        for o in ops:
            if o.startswith('v_mfma'):
                k = 'mfma'
            elif o.startswith('v_pk'):
                k = 'v_pk'
            elif o.startswith(('v_readlane', 'v_writelane')):
                k = 'lane(spill)'
            elif o.startswith('v_'):
                k = 'valu:' + o
            elif o.startswith(('ds_', 'buffer_', 'global_')):
                k = o
            elif o.startswith(('s_waitcnt', 's_nop', 's_barrier')):
                k = o
            elif o.startswith('s_'):
                k = 'salu'
            else:
                k = o
            c[k] += 1
        print(name, len(ops), dict(c))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
