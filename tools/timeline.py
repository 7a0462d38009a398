"""Decoder-loop timeline from a rocprofv3 kernel trace (development tool).
    python tools/timeline.py <kernel_trace.csv> fwd|bwd [first_step] [n_rows]
Prints the span and the per-step period of the last loop of that kind in the trace (anchor kernel: the energies kernel of
the attention chain), then `n_rows` kernels from step `first_step` on with queue, workgroups, start, duration and the gap
to the previous kernel on the same queue.  Under the profiler the host issues launches more slowly than the GPU retires
them in the two-stream schedule, so gaps on a queue are not representative; kernel durations are."""
import collections, csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mode = sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 400
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
names = (('rec_cell_kernel', 'RC'), ('rec_gemm_kernel', 'G'), ('cell_query_kernel', 'CQ'), ('cell_bwd_query', 'CBQ'),
         ('attn_energies_fwd', 'Ke'), ('attn_context_fwd', 'Ks'), ('attn_context_bwd', 'Ksb'), ('attn_energies_bwd', 'Keb'),
         ('lstm_cell_fwd', 'C'), ('lstm_cell_bwd', 'Cb'), ('sgemm', 'MM'), ('attn_param', 'PG'))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['k'] = next((s for key, s in names if key in r['Kernel_Name']), 'o')
rows.sort(key=lambda r: r['s'])
anchor = [i for i, r in enumerate(rows) if r['k'] == ('Ke' if mode == 'fwd' else 'Keb')][-870:]
gaps = [(rows[anchor[i + 1]]['s'] - rows[anchor[i]]['s']) / 1e3 for i in range(len(anchor) - 1)]
print(f'loop span {(rows[anchor[-1]]["s"] - rows[anchor[0]]["s"]) / 1e6:.2f} ms, step period median {statistics.median(gaps):.2f} us, '
      f'mean {sum(gaps) / len(gaps):.2f} us')
dur = collections.defaultdict(list)
for r in rows[anchor[0]:anchor[-1]]:
    dur[(r['k'], r['Queue_Id'])].append((r['e'] - r['s']) / 1e3)
print('  '.join(f'{k}@q{q}: {statistics.median(v):.1f} us x{len(v)}' for (k, q), v in sorted(dur.items()) if len(v) > 50))
i0 = anchor[first]
t0 = rows[i0]['s']
prev_end = {}
for r in rows[i0:i0 + n]:
    q = r['Queue_Id']
    gap = (r['s'] - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = r['e']
    wgs = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * \
        (int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    print(f"{r['k']:4s} q{q:>2s} wgs {wgs:5d}  start {(r['s'] - t0) / 1e3:8.2f}  dur {(r['e'] - r['s']) / 1e3:6.2f}  gap-on-queue {gap:6.2f}")
