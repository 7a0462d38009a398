#!/bin/bash
# Run ON THE GPU BOX (gpurun): the bench lines kept under profiles/rNN_*_bench_line.json, one box, one after the other.
#   bash tools/collect_lines.sh gpurun_out/r04lines
set -o pipefail
OUT=${1:-gpurun_out/lines}
mkdir -p "$OUT"
line() { name=$1; shift; python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" && echo "$name: $(python -c "import json,sys; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms', round(d['value']), d['unit'])")"; }
line default --steps 20 --warmup 5 &&
line bf16 --no-cpu-baseline --bf16 --steps 16 --warmup 6 &&
line vesus --no-cpu-baseline --vesus --steps 16 --warmup 6 &&
line merged_b128_bf16 --no-cpu-baseline --merged --steps 12 --warmup 6 &&
line b128 --no-cpu-baseline --batch 128 --steps 5 --warmup 2 &&
line planes3 --no-cpu-baseline --fp32-split 3 --steps 16 --warmup 6 &&
line fp32pipe --no-cpu-baseline --fp32-split 0 --steps 16 --warmup 6 &&
GANTRON_SCHEDULE=no_attn_bwd_cell line schedule_no_attn_bwd_cell --no-cpu-baseline --no-data-path --steps 16 &&
GANTRON_EXPERIMENTAL=plstm=7 line experimental_plstm7 --no-cpu-baseline --no-data-path --steps 16 &&
GANTRON_SCHEDULE=no_plstm line schedule_no_plstm --no-cpu-baseline --no-data-path --steps 16 &&
GANTRON_SCHEDULE=no_blk_gemm_aside line schedule_no_blk_gemm_aside --no-cpu-baseline --no-data-path --steps 16 &&
GANTRON_SCHEDULE=no_rec_h2 line schedule_no_rec_h2 --no-cpu-baseline --no-data-path --steps 16 &&
line same_box_default --no-cpu-baseline --no-data-path --steps 16
