import sys, os
sys.path[:0] = ['.', 'tests', 'tests/golden']
import torch
import test_parity_gpu as T
from gantron_amd import engine
dev = torch.device('cuda:0')
for i in range(3):
    try:
        T.test_full_size_batch_is_consistent_with_its_sub_batch(dev)
        print('run', i, 'OK')
    except AssertionError as e:
        print('run', i, 'FAILED', str(e)[:120])
    try:
        engine.check_handoffs()
        print('  handoffs clean')
    except Exception as e:
        print('  handoff flag:', e)
