run() { python tools/sweep_attr.py "$@" -- --steps 8 --warmup 2 --no-cpu-baseline --no-data-path 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%.2f ms  g %.2f d %.2f' % (d['ms_per_step'], d['config']['g_step_ms'], d['config']['d_step_ms']))"; }
echo "base:"; run
echo "base again:"; run
for v in 768 1024; do echo "SPLITK_EXPOSED=$v:"; run ops.SPLITK_EXPOSED=$v ops.SPLITK_TARGET=$v; done
for v in 64 192 256 384; do echo "SPLITK_OVERLAPPED=$v:"; run ops.SPLITK_OVERLAPPED=$v; done
for v in 8 16 24; do echo "BWD_CHUNKS=$v:"; run engine.BWD_CHUNKS=$v; done
for v in 16 32; do echo "DEC_BLK_BWD=$v:"; run engine.DEC_BLK_BWD=$v; done
for v in 8 24 32; do echo "DEC_BLK=$v:"; run engine.DEC_BLK=$v; done
