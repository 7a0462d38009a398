"""Development tool: bench.py with ops.SPLITK_OVERLAPPED / SPLITK_EXPOSED overridden (OVL=<n> EXP=<n>) or with knock-outs
(NODW=1, NOAMAX=1), prints ms per pair."""
import json, os, runpy, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gantron_amd.ops as ops
if os.environ.get('OVL'):
    ops.SPLITK_OVERLAPPED = int(os.environ['OVL'])
if os.environ.get('EXP'):
    ops.SPLITK_EXPOSED = ops.SPLITK_TARGET = int(os.environ['EXP'])
if os.environ.get('CHUNKS'):
    import gantron_amd.engine as _eng
    _eng.BWD_CHUNKS = int(os.environ['CHUNKS'])
if os.environ.get('NODW'):       # knock-out: linear weight gradients not computed at all (what do they cost the pair?)
    ops.linear_dw = lambda *a, **k: None
if os.environ.get('NOAMAX'):     # knock-out: no shared-maximum passes (every product finds its own)
    ops.h2_active = lambda: False
sys.argv = ['bench.py', '--steps', os.environ.get('STEPS', '12'), '--warmup', '3', '--no-cpu-baseline', '--no-data-path']
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
line = json.loads(buf.getvalue().strip().splitlines()[-1])
print(f"OVL={os.environ.get('OVL')} EXP={os.environ.get('EXP')} DEBUG={os.environ.get('GANTRON_DEBUG')} NODW={os.environ.get('NODW')} CHUNKS={os.environ.get('CHUNKS')}: {line['ms_per_step']:.2f} ms  pairs {line['config'].get('pair_ms_min_median_max')}", flush=True)
