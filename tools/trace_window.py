"""Kernels of a time window of a rocprofv3 kernel trace (development tool).
    python tools/trace_window.py <kernel_trace.csv> <anchor substring> <nth occurrence> [n_rows]
Prints n_rows kernels starting at the nth launch whose name contains the anchor: queue, workgroups, start, duration, gap to
the previous kernel on the same queue; then the median duration of every kernel name in the trace."""
import collections, csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor, nth = sys.argv[2], int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['n'] = re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')))[:44]
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
print(f'{len(idx)} launches match {anchor!r}')
i0 = idx[nth]
t0 = rows[i0]['s']
prev_end = {}
for r in rows[max(0, i0 - 200):i0]:
    prev_end[r['Queue_Id']] = r['e']
for r in rows[i0:i0 + n]:
    q = r['Queue_Id']
    gap = (r['s'] - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = r['e']
    wgs = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * \
        (int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    print(f"{r['n']:44s} q{q:>2s} wgs {wgs:5d}  start {(r['s'] - t0) / 1e3:9.2f}  dur {(r['e'] - r['s']) / 1e3:7.2f}  gap {gap:7.2f}")
dur = collections.defaultdict(list)
for r in rows:
    dur[r['n']].append((r['e'] - r['s']) / 1e3)
print()
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f'{k:44s} x{len(v):6d}  median {statistics.median(v):8.2f} us  total {sum(v) / 1e3:8.2f} ms')
