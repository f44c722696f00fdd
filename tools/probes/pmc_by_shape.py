"""Per (kernel, grid) sum of one PMC counter of a rocprofv3 counter_collection.csv (KiB counters -> MB): which launch shapes carry a kernel's traffic.
    python tools/probes/pmc_by_shape.py <run_counter_collection.csv> <COUNTER> <steps>"""
import collections, csv, sys
sys.path.insert(0, 'tools')
from kernel_table import short
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] != sys.argv[2]:
        continue
    k = (short(r['Kernel_Name']), int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1))
    acc[k][0] += 1
    acc[k][1] += float(r['Counter_Value']) * 1024 / 1e6
steps = float(sys.argv[3])
for (name, grid), (n, mb) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if mb / steps > 50:
        print('%-62s %8d %6.1f %9.1f MB/launch %9.1f MB/step' % (name[:62], grid, n / steps, mb / n, mb / steps))
