"""Build libgantron_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m gantron_amd.csrc.build        # or: python gantron_amd/csrc/build.py
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ['api.hip', 'gemm.hip', 'recurrent.hip', 'recurrent_bf16.hip', 'recurrent_fused.hip', 'cell_query.hip', 'attention.hip', 'attention_mfma.hip', 'elementwise.hip', 'loss_optim.hip', 'loops.hip', 'plstm.hip', 'collective.hip']
LIB = os.path.join(PKG, 'libgantron_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-fno-gpu-rdc', '-DNDEBUG']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    # every header next to the sources holds device code of some kernel (rec_cell.h, attn_ctx_bwd.h, ...): any of them
    # newer than an object rebuilds it
    headers = sorted(glob.glob(os.path.join(HERE, '*.h'))) + [os.path.join(PKG, '..', 'include', 'gantron_hip.h')]
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, '-x', 'hip', '-c', s, '-o', o] + FLAGS)

    def run(cmd):
        if verbose:
            print(' '.join(cmd[:6]), '...', flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs + ['-ldl'])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
