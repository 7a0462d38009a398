#!/bin/bash
# Feasibility probe (round 3): is the chip idle enough beside ONE latency-bound decoder chain that two independent half
# batches, each with its own chains, finish sooner than one full batch?  Runs bench.py --batch 32 alone, then two of them at
# the same time (two processes, one GPU), and prints ms per pair of each.
set -e
mkdir -p gpurun_out/halves
A="--steps 8 --warmup 3 --no-data-path --no-cpu-baseline"
python bench.py --batch 64 $A 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one process  B=64:', round(l['ms_per_step'],2), 'ms per pair', round(l['value']), 'frames/s')"
python bench.py --batch 32 $A 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one process  B=32:', round(l['ms_per_step'],2), 'ms per pair', round(l['value']), 'frames/s')"
python bench.py --batch 32 $A > gpurun_out/halves/a.json 2>/dev/null &
PA=$!
python bench.py --batch 32 $A > gpurun_out/halves/b.json 2>/dev/null &
PB=$!
wait $PA; wait $PB
for f in a b; do python -c "import sys,json; l=json.loads(open('gpurun_out/halves/$f.json').read().strip().splitlines()[-1]); print('two processes B=32 ($f):', round(l['ms_per_step'],2), 'ms per pair', round(l['value']), 'frames/s')"; done
