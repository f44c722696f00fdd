"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter values over the dispatches of each kernel name
(the first dispatch of a name is dropped as warm-up when there are several)."""
import collections, csv, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        rows[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for k, cs in rows.items():
    if not any(s in k for s in ('conv', 'gemm', 'wino')):
        continue
    out = {}
    dur = None
    for c, v in cs.items():
        v = sorted(v)[1:] if len(v) > 1 else v
        out[c] = sum(x[1] for x in v) / len(v)
        dur = sum(x[2] for x in v) / len(v)
    print(k, 'dur_us=%.1f' % (dur / 1e3))
    for c in sorted(out):
        print('   %-28s %.4g' % (c, out[c]))
    g = out.get('GRBM_GUI_ACTIVE')
    if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in out:
        print('   mfma busy / (4 SIMD x 256 CU x GUI_ACTIVE) = %.3f   clock = %.2f GHz' % (out['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * g), g / dur))
    if 'SQ_INSTS_MFMA' in out and 'SQ_INSTS_VALU' in out:
        print('   VALU (non-MFMA) per MFMA = %.2f' % ((out['SQ_INSTS_VALU'] - out['SQ_INSTS_MFMA']) / out['SQ_INSTS_MFMA']))
