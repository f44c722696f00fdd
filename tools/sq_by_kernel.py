"""Sum a rocprofv3 --pmc counter_collection.csv per kernel name (one bench step with --steps 1 --warmup 1 = 2 steps of dispatches):
    python tools/sq_by_kernel.py <run_counter_collection.csv> <dst.csv>"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    for ns in ('l2i_h8_bf16::', 'l2i_h8s_bf16::', 'l2i_h8_f16::', 'l2i_h8s_f16::'):
        k = k.replace(ns, '')
    rows[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r['Dispatch_Id'])
names = sorted({c for v in rows.values() for c in v})
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Kernel_Name', 'Dispatches(2 steps)'] + names)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)):
        w.writerow([k, len(disp[k])] + ['%.6g' % v.get(c, 0.0) for c in names])
