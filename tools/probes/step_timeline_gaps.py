"""[r6] From a rocprofv3 kernel trace of hipGraph-replayed steps: for the LAST replayed step, the wall span, the time covered by kernels of >= 15 us (union over
streams), the time covered only by shorter kernels, and the idle time.   usage: python tools/probes/step_timeline_gaps.py <kernel_trace.csv> <ms_per_step>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
step_ms = float(sys.argv[2])
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:]) for r in rows), key=lambda e: e[0])
t_end = ev[-1][1]
t0 = t_end - int(step_ms * 1e6 * 1.0)
win = [e for e in ev if e[0] >= t0]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


big = [(s, e) for s, e, n in win if e - s >= 15000]
allk = [(s, e) for s, e, n in win]
span = win[-1][1] - win[0][0]
print('window %.2f ms: %d kernels, %d of them >= 15 us' % (span / 1e6, len(win), len(big)))
print('covered by >= 15 us kernels %.2f ms; by any kernel %.2f ms; idle %.2f ms; only-small %.2f ms' %
      (union(big) / 1e6, union(allk) / 1e6, (span - union(allk)) / 1e6, (union(allk) - union(big)) / 1e6))
# the longest stretches without a big kernel
bs = sorted(big)
gaps = []
cur = win[0][0]
for s, e in bs:
    if s > cur:
        inside = [n for ss, ee, n in win if ss >= cur and ee <= s]
        gaps.append((s - cur, cur - win[0][0], len(inside), inside[:6]))
    cur = max(cur, e)
gaps.sort(reverse=True)
for g in gaps[:12]:
    print('  gap %.1f us at +%.2f ms: %d small kernels, first: %s' % (g[0] / 1e3, g[1] / 1e6, g[2], [n.split('::')[-1][:40] for n in g[3]]))
