"""Per (kernel, grid size) table of a rocprofv3 --kernel-trace CSV: which LAUNCH SHAPES of a kernel carry its time.

    python tools/trace_by_shape.py <run_kernel_trace.csv> <steps in the file> [min ms per step] > table.txt

columns: kernel, grid (workgroups), launches per step, mean microseconds per launch, ms per step."""
import collections
import csv
import sys

from kernel_table import short


def main(path, steps, min_ms=0.1):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        wg = max(int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1), 1)
        grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) // wg
        gy = int(r.get('Grid_Size_Y', 1) or 1) // max(int(r.get('Workgroup_Size_Y', 1) or 1), 1)
        gz = int(r.get('Grid_Size_Z', 1) or 1) // max(int(r.get('Workgroup_Size_Z', 1) or 1), 1)
        k = (short(r['Kernel_Name']), grid * gy * gz)
        acc[k][0] += 1
        acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    print('%-64s %9s %8s %10s %9s' % ('kernel', 'blocks', 'n/step', 'us/launch', 'ms/step'))
    for (name, grid), (n, us) in rows:
        if us / steps / 1e3 < min_ms:
            continue
        print('%-64s %9d %8.1f %10.1f %9.3f' % (name[:64], grid, n / steps, us / n, us / steps / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0.1)
