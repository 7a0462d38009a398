"""One training process per visible GPU (reference multiproc.py:1-23): `python -m gantron_amd.multiproc
-m gantron_amd.train <train args>`; ranks > 0 log to logs/<group>_GPU_<i>.log."""
import os
import subprocess
import sys
import time

import torch


def main():
    argslist = list(sys.argv)[1:]
    num_gpus = torch.cuda.device_count()
    argslist.append('--n_gpus={}'.format(num_gpus))
    job_id = time.strftime('%Y_%m_%d-%H%M%S')
    argslist.append('--group_name=group_{}'.format(job_id))
    os.makedirs('logs', exist_ok=True)
    workers = []
    for i in range(num_gpus):
        argslist.append('--rank={}'.format(i))
        stdout = None if i == 0 else open('logs/{}_GPU_{}.log'.format(job_id, i), 'w')
        workers.append(subprocess.Popen([sys.executable] + argslist, stdout=stdout))
        argslist = argslist[:-1]
    for p in workers:
        p.wait()


if __name__ == '__main__':
    main()
