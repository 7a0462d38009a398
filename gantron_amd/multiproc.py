"""Single-node launcher: one training process per MI355X, rendezvous through the environment.

    python -m gantron_amd.multiproc [--gpus N] [--log-dir logs] -m gantron_amd.train <train arguments>
    python -m gantron_amd.multiproc path/to/script.py <arguments>

Stands in for the reference's multiproc.py:1-23 (one `train.py --n_gpus --rank --group_name` copy per visible GPU),
re-designed for one-process-per-GPU RCCL jobs:

* every worker gets the torchrun environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, a free
  MASTER_PORT) -- `gantron_amd.distributed.init_distributed` prefers it over hparams.dist_url, and bench.py reads
  nothing else -- plus the reference's `--n_gpus / --rank / --group_name` flags when the target is a training script
  that takes them (`--no-rank-flags` leaves the command line untouched);
* HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, required by RCCL on this driver) and GPU_MAX_HW_QUEUES=8 (the step drives
  three HIP streams next to RCCL's) are set for the workers;
* the launcher supervises: when one rank exits with an error the others are terminated instead of hanging in a
  collective forever (the reference just wait()s), and the launcher's exit status is the first failure's.
Rank 0 keeps the terminal; the other ranks write to <log-dir>/<group>_rank<i>.log.
"""
import argparse
import os
import signal
import socket
import subprocess
import sys
import time


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _visible_gpus():
    import torch
    return torch.cuda.device_count()      # counts devices without initialising the HIP runtime in this process


def worker_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('GPU_MAX_HW_QUEUES', '8')
    return env


def supervise(procs, poll_s=0.5, grace_s=10.0):
    """Wait for every worker; on the first failure terminate the rest.  Returns the exit status to propagate."""
    status = 0
    live = dict(enumerate(procs))
    while live:
        for rank, p in list(live.items()):
            rc = p.poll()
            if rc is None:
                continue
            del live[rank]
            if rc != 0 and status == 0:
                status = rc
                print(f'[multiproc] rank {rank} exited with status {rc}: stopping the other ranks', file=sys.stderr, flush=True)
                for q in live.values():
                    q.send_signal(signal.SIGTERM)
                deadline = time.time() + grace_s
                for q in live.values():
                    try:
                        q.wait(timeout=max(0.0, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
        if live:
            time.sleep(poll_s)
    return status


_OWN_FLAGS = {'--gpus': 1, '--log-dir': 1, '--no-rank-flags': 0, '-h': 0, '--help': 0}


def split_command_line(argv):
    """The launcher's own options end at the first token that is not one of them (`-m`, a script path, `--`): everything
    from there on is the target's command line, passed through untouched.  (argparse.REMAINDER cannot do this: it
    refuses an option-like first token such as `-m`.)"""
    i = 0
    while i < len(argv):
        key = argv[i].split('=', 1)[0]
        if key not in _OWN_FLAGS:
            break
        i += 1 + (_OWN_FLAGS[key] if '=' not in argv[i] else 0)
    own, target = argv[:i], argv[i:]
    if target[:1] == ['--']:
        target = target[1:]
    return own, target


def main(argv=None):
    ap = argparse.ArgumentParser(prog='gantron_amd.multiproc', description=__doc__.split('\n')[0],
                                 usage='%(prog)s [--gpus N] [--log-dir DIR] [--no-rank-flags] (-m module | script.py) [arguments ...]')
    ap.add_argument('--gpus', type=int, default=0, help='processes to start (default: every visible GPU)')
    ap.add_argument('--log-dir', default='logs')
    ap.add_argument('--no-rank-flags', action='store_true',
                    help='do not append --n_gpus/--rank/--group_name (the target reads the environment only)')
    own, target = split_command_line(list(sys.argv[1:] if argv is None else argv))
    args = ap.parse_args(own)
    args.target = target
    if not args.target:
        ap.error('nothing to launch')
    world = args.gpus or _visible_gpus()
    if world < 1:
        ap.error('no GPU visible and --gpus not given')
    group = time.strftime('group_%Y_%m_%d-%H%M%S')
    port = _free_port()
    os.makedirs(args.log_dir, exist_ok=True)
    procs, logs = [], []
    for rank in range(world):
        cmd = [sys.executable] + list(args.target)
        if not args.no_rank_flags:
            cmd += [f'--n_gpus={world}', f'--rank={rank}', f'--group_name={group}']
        out = None
        if rank > 0:
            out = open(os.path.join(args.log_dir, f'{group}_rank{rank}.log'), 'w')
            logs.append(out)
        procs.append(subprocess.Popen(cmd, env=worker_env(rank, world, port), stdout=out,
                                      stderr=subprocess.STDOUT if out else None))
    try:
        return supervise(procs)
    except KeyboardInterrupt:
        for p in procs:
            p.send_signal(signal.SIGINT)
        return 130
    finally:
        for f in logs:
            f.close()


if __name__ == '__main__':
    sys.exit(main())
