# round 4: the fp16-plane recurrent products in the step -- forward only / backward only / both, and the backward product's K-splits
run() { python tools/sweep_attr.py "$@" -- --steps 16 --warmup 3 --no-cpu-baseline --no-data-path 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%.2f ms  pairs %s  g %.2f d %.2f' % (d['ms_per_step'], d['config'].get('pair_ms_min_median_max'), d['config']['g_step_ms'], d['config']['d_step_ms']))"; }
for v in 0 1 2 3 1 0; do echo "REC_H2=$v:"; run engine.REC_H2=$v; done
for v in 4 8 12; do echo "REC_H2=3 S1_BWD=$v:"; run engine.REC_H2=3 engine.S1_BWD=$v; done
for v in 2 4; do echo "REC_H2=1 S1_FWD... (h2 forward splits via rec_splits_x3 cus) skipped"; break; done
