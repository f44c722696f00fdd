"""Per-kernel table of one bench step from committed / collected rocprofv3 outputs:

    python tools/kernel_table.py profiles/r01_bench_serial_streams_kernel_stats.csv profiles/r01_conv_hbm_traffic.json \
        profiles/r01_step_sq_counters_by_kernel.csv profiles/r01_kernel_table.md

columns: launches and ms per step (rocprofv3 --kernel-trace --stats of bench.py --serial_streams, 12 steps), HBM GB per step and
TB/s (FETCH_SIZE x 2 + WRITE_SIZE passes, tools/hbm_traffic.py), matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8 XCDs) and non-MFMA VALU instructions per MFMA (one SQ pass over one step)."""
import collections
import csv
import json
import sys


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    for ns in ('l2i_h8_bf16::', 'l2i_h8s_bf16::', 'l2i_h8_f16::', 'l2i_h8s_f16::', 'l2i_pair_f32::'):      # [r5] per-element-type namespaces of the h8 kernels
        n = n.replace(ns, '')
    return n


def main(stats_csv, traffic_json, sq_csv, dst, steps_in_stats=12):
    stats = {}
    for r in csv.DictReader(open(stats_csv)):
        stats[short(r['Name'])] = (int(r['Calls']) / steps_in_stats, float(r['TotalDurationNs']) / steps_in_stats / 1e6)
    traffic = json.load(open(traffic_json))['per_kernel']
    sq = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sq_csv)):
        if 'Counter_Name' in r:                       # raw rocprofv3 counter_collection.csv: one row per dispatch and counter
            sq[short(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
        else:                                         # per-kernel sums (profiles/r01_step_sq_counters_by_kernel.csv)
            for k, v in r.items():
                if k not in ('Kernel_Name', 'Dispatches(2 steps)'):
                    sq[short(r['Kernel_Name'])][k] += float(v)
    lines = ['| kernel | launches/step | ms/step | HBM GB/step | TB/s | matrix-pipe busy | VALU per MFMA |', '|---|---|---|---|---|---|---|']
    tot = sum(v[1] for v in stats.values())
    for k, (n, ms) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if ms < 0.3:
            continue
        t = traffic.get(k, {})
        gb = t.get('read_GB_per_step', 0.0) + t.get('write_GB_per_step', 0.0)
        c = sq.get(k, {})
        busy = vpm = ''
        if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_INSTS_MFMA'):
            busy = '%.0f %%' % (100.0 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0))
            vpm = '%.2f' % ((c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA'])
        lines.append('| `%s` | %.1f | %.2f | %s | %s | %s | %s |' % (k if len(k) <= 72 else k[:69] + '...', n, ms, ('%.1f' % gb) if gb else '', ('%.2f' % (gb / ms)) if gb else '', busy, vpm))
    lines.append('')
    lines.append('Total kernel time %.1f ms per step (kernels under 0.3 ms per step omitted from the table).' % tot)
    open(dst, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main(*sys.argv[1:5], **({'steps_in_stats': int(sys.argv[5])} if len(sys.argv) > 5 else {}))
