# round 4, after the two-launch backward step + the persistent decoder-LSTM backward: do the loop's tuned constants still hold?
run() { python tools/sweep_attr.py "$@" -- --steps 16 --warmup 3 --no-cpu-baseline --no-data-path 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%.2f ms  pairs %s  g %.2f d %.2f' % (d['ms_per_step'], d['config'].get('pair_ms_min_median_max'), d['config']['g_step_ms'], d['config']['d_step_ms']))"; }
echo "base:"; run
for v in 16 32; do echo "DEC_BLK_BWD=$v:"; run engine.DEC_BLK_BWD=$v; done
echo "base:"; run
for v in 8 16; do echo "BWD_CHUNKS=$v:"; run engine.BWD_CHUNKS=$v; done
for v in 256 512; do echo "SPLITK_OVERLAPPED=$v:"; run ops.SPLITK_OVERLAPPED=$v; done
echo "base:"; run
for v in 16 32; do echo "DEC_BLK=$v:"; run engine.DEC_BLK=$v; done
