"""Probe (round 4): host time of the pieces between the encoder convolutions and the BiLSTM of a generator forward (a kernel
trace showed the GPU idle there), measured inside ordinary bench steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gantron_amd import engine, ops
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc.setdefault(name, []).append(1e3 * (time.perf_counter() - t)); return r
    return w
engine.GeneratorEngine.step_major_index = staticmethod(timed('step_major_index', engine.GeneratorEngine.step_major_index))
ops.h2d = timed('h2d', ops.h2d)
engine.ops.h2d = ops.h2d
engine.GeneratorEngine._encode = timed('_encode (host)', engine.GeneratorEngine._encode)
engine.GeneratorEngine.forward = timed('forward (host)', engine.GeneratorEngine.forward)
engine.GeneratorEngine.backward = timed('backward (host)', engine.GeneratorEngine.backward)
import bench
sys.argv = ['bench.py', '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-data-path']
bench.main()
for k, v in acc.items():
    v = v[len(v) // 3:]
    print(f'{k:22s} calls {len(v):4d}  mean {sum(v) / len(v):8.3f} ms  max {max(v):8.3f} ms', file=sys.stderr)
