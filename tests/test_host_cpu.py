"""CPU-only checks: the C-ABI library loads and exports what the header declares, host logic
(collate, text front-end, packed-sequence index maps, flat parameters), and the data-parallel
gradient averaging over gloo with world_size 2.  No kernel is launched here."""
import ctypes
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gantron_amd import _lib
    lib = _lib.lib()                       # raises on a missing symbol or a struct-size mismatch
    assert lib.gt_version() >= 100
    declared = set(_lib.PROTOS)
    header = open(_lib.HEADER).read()
    in_header = set(re.findall(r'\b(gt_[a-z0-9_]+)\s*\(', header))
    assert in_header == declared, in_header ^ declared
    for name in declared:
        assert hasattr(lib, name)
    # argument validation happens before any launch: a bad call must fail cleanly without a GPU
    rc = lib.gt_sgemm(None, None)
    assert rc == -1 and b'NULL' in lib.gt_last_error_string()


def test_product_path_never_touches_the_oracle():
    """The shipped package must not import, load or execute anything under oracle/."""
    pat = re.compile(r'^\s*(from|import)\s+oracle\b|oracle[/\\.]gantron_ref|[\'"]oracle[\'"]', re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'gantron_amd')):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f'{f} references the oracle package'


def test_module_state_dict_matches_reference_golden(golden_dir):
    import common
    from gantron_amd.hparams import HParams
    from gantron_amd.model import Discriminator, Tacotron2
    z = np.load(os.path.join(golden_dir, 'full_pair_summary.npz'))
    shapes_g = {k[len('shape_g/'):]: tuple(z[k]) for k in z.files if k.startswith('shape_g/')}
    shapes_d = {k[len('shape_d/'):]: tuple(z[k]) for k in z.files if k.startswith('shape_d/')}
    hp = HParams(common.FULL_HPARAMS)
    G, D = Tacotron2(hp), Discriminator(hp)
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == shapes_g
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == shapes_d
    assert sum(p.numel() for p in G.parameters()) == 32494465
    assert sum(p.numel() for p in D.parameters()) == 12331169


def test_collate_matches_reference_semantics():
    from gantron_amd.data_utils import TextMelCollate
    g = torch.Generator().manual_seed(0)
    items = [(torch.randint(1, 100, (n,), generator=g).int(), torch.randn(80, t, generator=g))
             for n, t in ((5, 30), (9, 44), (7, 12))]
    text, il, mel, gate, spk, emo, ol = TextMelCollate(1)(items)
    assert il.tolist() == [9, 7, 5] and ol.tolist() == [44, 12, 30]
    assert text.shape == (3, 9) and mel.shape == (3, 80, 44) and gate.shape == (3, 44)
    assert torch.equal(text[1, :7], items[2][0].long()) and text[1, 7:].abs().sum() == 0
    assert torch.equal(mel[2, :, :30], items[0][1]) and mel[2, :, 30:].abs().sum() == 0
    assert gate[1, :11].sum() == 0 and gate[1, 11:].min() == 1 and gate[0, 43] == 1 and gate[0, :43].sum() == 0


def test_text_frontend():
    from gantron_amd.text import sequence_to_text, text_to_sequence
    from gantron_amd.text.symbols import symbols
    assert len(symbols) == 148 and symbols[0] == '_' and symbols[1] == '-'
    seq = text_to_sequence('Dr. Smith owes $2 to 1,000 people since 1984. {HH AW1}', ['english_cleaners'])
    assert sequence_to_text(seq) == 'doctor smith owes two dollars to one thousand people since nineteen eighty-four. {HH AW1}'
    assert all(0 < s < 148 for s in seq)


def test_step_major_index_roundtrip():
    from gantron_amd.engine import GeneratorEngine
    lens, T, Bp = [6, 4, 1], 6, 16
    fwd, rev, ifwd, irev, active = GeneratorEngine.step_major_index(lens, T, Bp)
    x = torch.arange(len(lens) * T, dtype=torch.float32)
    for d, (m, im) in enumerate(((fwd, ifwd), (rev, irev))):
        xs = torch.where(m >= 0, x[m.clamp(min=0).long()], torch.zeros(()))
        back = torch.where(im >= 0, xs[im.clamp(min=0).long()], torch.full((), -1.0))
        for b, L in enumerate(lens):
            assert torch.equal(back[b * T:b * T + L], x[b * T:b * T + L])
            assert (back[b * T + L:(b + 1) * T] == -1).all()
    assert active[:, 0].tolist() == [1] * 6 and active[:, 2].tolist() == [1, 0, 0, 0, 0, 0] and active[:, 3:].sum() == 0
    assert rev.view(T, Bp)[0, 1] == 1 * T + 3          # reverse direction starts at the last valid position


def test_flatten_module_keeps_values_and_shares_storage():
    from gantron_amd.optim import flatten_module
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat, grad = flatten_module(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in m.parameters())
    assert flat.numel() % 4 == 0 and grad.shape == flat.shape


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[3])
from gantron_amd.distributed import apply_gradient_allreduce
rank, port = int(sys.argv[1]), sys.argv[2]
dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', world_size=2, rank=rank)
torch.manual_seed(100 + rank)
lin = torch.nn.Linear(4, 3)
lin = apply_gradient_allreduce(lin)
w = lin.weight.detach().clone().numpy()
x = torch.full((2, 4), float(rank + 1))
lin(x).sum().backward()
np.savez(sys.argv[4] + f'/rank{rank}.npz', w=w, gw=lin.weight.grad.numpy(), gb=lin.bias.grad.numpy())
dist.destroy_process_group()
'''


def test_gradient_allreduce_two_ranks_gloo(tmp_path, golden_dir):
    """Same experiment as the reference fixture (tests/golden/make_golden.py:allreduce_fixture): after the
    wrap both ranks hold rank 0's weights, after backward both hold the mean gradient."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '29581', ROOT, str(tmp_path)], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    z = np.load(os.path.join(golden_dir, 'allreduce.npz'))
    for r in range(2):
        got = np.load(tmp_path / f'rank{r}.npz')
        np.testing.assert_allclose(got['w'], z[f'w_after_bcast_{r}'], rtol=0, atol=0)
        np.testing.assert_allclose(got['gw'], z[f'grad_w_{r}'], rtol=1e-6)
        np.testing.assert_allclose(got['gb'], z[f'grad_b_{r}'], rtol=1e-6)
    np.testing.assert_allclose(np.load(tmp_path / 'rank0.npz')['gw'], np.full((3, 4), 3.0))


def test_bench_helpers():
    sys.path.insert(0, ROOT)
    import bench
    from gantron_amd.hparams import HParams
    hp = HParams(bench.HPARAMS)
    batch = bench.ljspeech_shaped_batch(hp, 64, seed=1234)
    il, ol = batch[1].tolist(), batch[6].tolist()
    assert il == sorted(il, reverse=True) and 12 <= min(il) and max(il) <= 187 and 96 <= min(ol) and max(ol) <= 870
    fixed = bench.ljspeech_shaped_batch(hp, 64, seed=0, fixed=True)
    fl = bench.pair_flops(hp, fixed[1].tolist(), fixed[6].tolist())
    assert abs(fl / 1e12 - 9.38) < 0.05          # 9.38 TFLOP per pair at B=64, 150/800
    # config 4 / 5 workloads: VESUS conditioning (E = 1093), mixed lengths, sorted by text length, labels only on non-LJ items
    hv = HParams(bench.VESUS_HPARAMS)
    for gen, B in ((bench.vesus_shaped_batch, 64), (bench.merged_shaped_batch, 128)):
        text, il, mel, gate, spk, emo, ol = gen(hv, B, seed=5)
        assert text.shape[0] == B and il.tolist() == sorted(il.tolist(), reverse=True)
        assert int(spk.max()) <= 122 and ((spk == 0) == (emo.sum(1) == 0)).all() and int((spk == 0).sum()) == B // 2
        assert all(float(gate[b, int(ol[b]) - 1]) == 1 and (int(ol[b]) < 2 or float(gate[b, int(ol[b]) - 2]) == 0) for b in range(B))
    fixed_v = bench.ljspeech_shaped_batch(hv, 64, seed=0, fixed=True)
    assert abs(bench.pair_flops(hv, fixed_v[1].tolist(), fixed_v[6].tolist()) / 1e12 - 9.56) < 0.05   # 9.56 at E=1093


def test_flat_adam_state_dict_uses_torch_adam_layout():
    """Checkpoints written by the reference hold torch.optim.Adam state dicts (train.py:158-166)."""
    from gantron_amd.optim import FlatAdam
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    ref = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-6)
    m(torch.randn(4, 5)).sum().backward()
    ref.step()
    opt = FlatAdam(m, lr=5e-4, weight_decay=1e-6)
    opt.load_state_dict(ref.state_dict())
    assert opt.step_count == 1 and opt.param_groups[0]['lr'] == 1e-3
    sd = opt.state_dict()
    rsd = ref.state_dict()
    assert sd['param_groups'][0]['params'] == rsd['param_groups'][0]['params']
    for i, st in rsd['state'].items():
        assert torch.allclose(sd['state'][i]['exp_avg'], st['exp_avg'])
        assert torch.allclose(sd['state'][i]['exp_avg_sq'], st['exp_avg_sq'])
        assert float(sd['state'][i]['step']) == float(st['step'])
    ref2 = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-6)
    ref2.load_state_dict(sd)                     # and back: the reference's optimizer accepts our dict


def test_multiproc_launcher_env_and_supervision(tmp_path):
    """gantron_amd.multiproc: every rank gets the torchrun environment plus the reference's --n_gpus/--rank/--group_name
    flags (reference multiproc.py:13-20), ranks > 0 log to files, and a failing rank stops the job instead of leaving
    the others hanging."""
    import subprocess
    script = tmp_path / 'worker.py'
    script.write_text(
        'import os, sys, time\n'
        'rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])\n'
        'assert f"--rank={rank}" in sys.argv and f"--n_gpus={world}" in sys.argv, sys.argv\n'
        'assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"\n'
        'print("worker", rank, "of", world, flush=True)\n'
        'if "--fail" in sys.argv and rank == 1:\n'
        '    sys.exit(7)\n'
        'if "--fail" in sys.argv:\n'
        '    time.sleep(60)\n')
    base = [sys.executable, '-m', 'gantron_amd.multiproc', '--gpus', '2', '--log-dir', str(tmp_path / 'logs'), str(script)]
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert 'worker 0 of 2' in r.stdout
    logs = list((tmp_path / 'logs').glob('*_rank1.log'))
    assert len(logs) == 1 and 'worker 1 of 2' in logs[0].read_text()
    t0 = time.time()
    r = subprocess.run(base + ['--fail'], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 7 and time.time() - t0 < 30, (r.returncode, r.stderr)
    # the module form the docstring and the README document: `multiproc [options] -m package.module <arguments>`
    pkg = tmp_path / 'wpkg'
    pkg.mkdir()
    (pkg / '__init__.py').write_text('')
    (pkg / 'job.py').write_text('from . import __name__ as pkg_name\n' + script.read_text() + 'assert "--extra=1" in sys.argv\n')
    for sep in ([], ['--']):
        cmd = [sys.executable, '-m', 'gantron_amd.multiproc', '--gpus=2', '--log-dir', str(tmp_path / 'logs_m')] + sep + \
              ['-m', 'wpkg.job', '--extra=1']
        r = subprocess.run(cmd, env=dict(env, PYTHONPATH=ROOT + os.pathsep + str(tmp_path)), capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and 'worker 0 of 2' in r.stdout, (r.returncode, r.stderr)


def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in front (no WORLD_SIZE): bench.py starts the two ranks itself and
    supervises them.  Without a GPU each rank stops at its "needs a GPU" check, so what is visible here is the launch
    itself: two children with RANK 0 / 1, a failing rank reported by the supervisor, its status propagated, no hang and
    no JSON line.  (The GPU box runs the same command to completion: tests/test_training_gpu.py.)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by test_bench_two_ranks_rehearsal')
    assert r.returncode != 0 and '[multiproc] rank' in r.stderr and 'needs a GPU' in r.stderr, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    # under a launcher whose world size contradicts --gpus the command refuses instead of asserting somewhere inside
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=dict(env, WORLD_SIZE='4', RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr


def test_multiproc_splits_its_own_options_from_the_target():
    from gantron_amd.multiproc import split_command_line
    assert split_command_line(['--gpus', '4', '-m', 'gantron_amd.train', '-o', 'out', '--gpus', '9']) == \
        (['--gpus', '4'], ['-m', 'gantron_amd.train', '-o', 'out', '--gpus', '9'])
    assert split_command_line(['--no-rank-flags', '--log-dir=x', 'bench.py', '--gpus', '2']) == \
        (['--no-rank-flags', '--log-dir=x'], ['bench.py', '--gpus', '2'])
    assert split_command_line(['--gpus', '2', '--', '-m', 'mod']) == (['--gpus', '2'], ['-m', 'mod'])
    assert split_command_line([]) == ([], [])


@pytest.mark.parametrize('tag', ['lj', 'vesus', 'vesus_multi'])
def test_loader_and_collate_match_the_reference(golden_dir, tmp_path, tag):
    """TextMelLoader (text front-end, seeded shuffle, mels from disk, VESUS speakers / labels) and TextMelCollate against
    loader_collate.npz, which tests/golden/make_golden.py produced by running the reference's own classes
    (data_utils.py:13-131) on the same small dataset (common.write_loader_dataset)."""
    import common
    from gantron_amd.data_utils import TextMelCollate, TextMelLoader
    from gantron_amd.hparams import HParams
    z = np.load(os.path.join(golden_dir, 'loader_collate.npz'))
    root = str(tmp_path)
    lj, ve = common.write_loader_dataset(root)
    extra = {'lj': '', 'vesus': f',vesus_path={root}/,use_intended_labels=True',
             'vesus_multi': f',vesus_path={root}/,use_intended_labels=False'}[tag]
    hp = HParams('use_labels=True,use_noise=True,n_mel_channels=8,load_mel_from_disk=True,seed=77' + extra)
    ds = TextMelLoader([lj, ve], hp, root + '/')
    assert len(ds) == int(z[f'{tag}/n'])
    assert list(ds.idx) == z[f'{tag}/idx'].tolist()
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        assert it[0].tolist() == z[f'{tag}/text{i}'].tolist(), f'text ids of item {i}'
        if tag != 'lj':
            assert int(it[2]) == int(z[f'{tag}/speaker{i}'])
            np.testing.assert_allclose(it[3].numpy(), z[f'{tag}/emotions{i}'])
    batch = TextMelCollate(1)(items[:5] if tag == 'lj' else items)
    names = ('text_padded', 'input_lengths', 'mel_padded', 'gate_padded', 'speaker_ids', 'emotions', 'output_lengths')
    for nme, t in zip(names, batch):
        key = f'{tag}/batch/{nme}'
        if key in z.files:
            assert tuple(t.shape) == z[key].shape and str(t.dtype).replace('torch.', '') == str(z[key].dtype), (nme, t.dtype, z[key].dtype)
            np.testing.assert_array_equal(t.numpy(), z[key], err_msg=nme)


def test_stft_refuses_to_run_without_the_gpu():
    """There is no CPU arithmetic path: TacotronSTFT on a CPU tensor (e.g. inside a forked DataLoader worker, where the
    reference runs its CPU STFT, data_utils.py:51-58) fails loudly and names the device data path instead."""
    from gantron_amd.stft import TacotronSTFT
    with pytest.raises(RuntimeError, match='DeviceBatches'):
        TacotronSTFT().mel_spectrogram(torch.zeros(1, 4096))


def test_load_wav_resamples_other_rates(tmp_path):
    """load_wav_to_torch (reference utils.py:27-31 = librosa.load(path, sr)): a 44.1 kHz and a 16 kHz file come back at
    22 050 Hz with the tone preserved (band-limited conversion; the filter itself is third-party, parity unpinned)."""
    from scipy.io import wavfile
    from gantron_amd.data_utils import load_wav_to_torch
    for sr in (44100, 16000, 22050):
        t = np.arange(sr) / sr                      # one second
        x = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
        path = str(tmp_path / f'tone_{sr}.wav')
        wavfile.write(path, sr, (x * 32767).astype(np.int16))
        y = load_wav_to_torch(path, 22050).numpy()
        assert abs(len(y) - 22050) <= 1 and y.dtype == np.float32
        ref = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(len(y)) / 22050)
        assert np.abs(y[2000:-2000] - ref[2000:-2000]).max() < 2e-3, sr


def test_switch_lists_reject_unknown_names_and_trim_like_the_c_parser(monkeypatch):
    """gantron_amd/switches.py: a mistyped switch raises with the list of valid names instead of silently running the default
    schedule; blanks around names, '=' and values are ignored (csrc/api.hip:gt_env_item follows the same rules -- checked
    here through the library, which needs no GPU for it)."""
    import ctypes
    import pytest
    from gantron_amd import switches as sw
    assert sw._parse('GANTRON_EXPERIMENTAL', ' plstm = 5 ; rec_tail ,graph=2') == {'plstm': 5, 'rec_tail': 1, 'graph': 2}
    with pytest.raises(ValueError, match='no_plstm'):
        sw._parse('GANTRON_SCHEDULE', 'no_pltsm')
    with pytest.raises(ValueError, match='decimal'):
        sw._parse('GANTRON_EXPERIMENTAL', 'plstm=0xf')
    with pytest.raises(AssertionError):
        sw.debug('no_such_diagnostic')
    from gantron_amd._lib import LIB_PATH
    h = ctypes.CDLL(LIB_PATH)
    fn = h._Z11gt_env_itemPKcS0_i if hasattr(h, '_Z11gt_env_itemPKcS0_i') else None
    if fn is None:
        pytest.skip('gt_env_item is not exported under its mangled name')
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    monkeypatch.setenv('GT_TEST_LIST', ' plstm = 5 ; rec_tail ,graph=2, no_blk_gemm_aside ,x')
    assert fn(b'GT_TEST_LIST', b'plstm', 0) == 5 and fn(b'GT_TEST_LIST', b'rec_tail', 0) == 1
    assert fn(b'GT_TEST_LIST', b'graph', 0) == 2 and fn(b'GT_TEST_LIST', b'no_blk_gemm_aside', 0) == 1
    assert fn(b'GT_TEST_LIST', b'rec', 7) == 7 and fn(b'GT_TEST_LIST', b'pl', 9) == 9        # prefixes are not matches
