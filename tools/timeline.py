"""Print a window of the decoder-loop timeline from a rocprofv3 kernel trace (development tool).
    python tools/timeline.py <kernel_trace.csv> fwd|bwd [first_step] [n_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mode = sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 400
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
names = (('rec_gemm_kernel', 'G'), ('cell_query_kernel', 'CQ'), ('cell_bwd_query', 'CBQ'), ('attn_energies_fwd', 'Ke'),
         ('attn_context_fwd', 'Ks'), ('attn_context_bwd', 'Ksb'), ('attn_energies_bwd', 'Keb'), ('lstm_cell_fwd', 'C'),
         ('lstm_cell_bwd', 'Cb'), ('sgemm', 'MM'))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['k'] = next((s for key, s in names if key in r['Kernel_Name']), 'o')
rows.sort(key=lambda r: r['s'])
anchor = [i for i, r in enumerate(rows) if r['k'] == ('CQ' if mode == 'fwd' else 'CBQ')][-870:]
i0 = anchor[first]
t0 = rows[i0]['s']
prev_end = {}
for r in rows[i0:i0 + n]:
    q = r['Queue_Id']
    gap = (r['s'] - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = r['e']
    wgs = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y'])))
    print(f"{r['k']:4s} q{q:>2s} wgs {wgs:5d}  start {(r['s'] - t0) / 1e3:8.2f}  dur {(r['e'] - r['s']) / 1e3:6.2f}  gap-on-queue {gap:6.2f}")
