#!/bin/bash
# Run ON THE GPU BOX (gpurun): every rocprofv3 pass behind profiles/rNN_* for the default bench workload.
#   bash tools/collect_profiles.sh gpurun_out/r02prof
# Counter passes are separate runs with --kernel-trace only (gpurun refuses --pmc together with the API trace domains).
set -o pipefail
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
# (round 3 ran the counter passes on ONE un-warmed pair: with more dispatches rocprofv3 7.2 segfaulted in its copy of a dispatch's
# kernel arguments, only for launches that went through hipExtLaunchKernelGGL with NULL events.  Since round 4 those launches are
# plain hipLaunchKernelGGL unless they are one of the profiler's samples (GT_LAUNCH_TIMED, csrc/common.h) and the passes run with
# the normal warm-up; PMC_ARGS overrides)
ARGS1="${PMC_ARGS:-bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-data-path}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o r04k -- python3 $ARGS > "$OUT/stats.log" 2>&1 && echo "stats pass done" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 $ARGS1 > "$OUT/fetch.log" 2>&1 && echo "fetch pass done" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 $ARGS1 > "$OUT/write.log" 2>&1 && echo "write pass done" &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/mfma" -o mfma -- python3 $ARGS1 > "$OUT/mfma.log" 2>&1 && echo "mfma pass done" &&
python3 tools/pmc_traffic.py "$OUT/fetch/fetch_counter_collection.csv" "$OUT/write/write_counter_collection.csv" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.txt" &&
python3 tools/pmc_mfma.py "$OUT/mfma/mfma_counter_collection.csv" "$OUT/pmc_mfma.json" > "$OUT/pmc_mfma.txt" &&
rm -f "$OUT"/fetch/*_counter_collection.csv "$OUT"/write/*_counter_collection.csv "$OUT"/mfma/*_counter_collection.csv "$OUT"/*/*_kernel_trace.csv &&
ls -la "$OUT" "$OUT/stats"
