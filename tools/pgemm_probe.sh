#!/bin/bash
# knock-outs of pgemm_kernel on the decoder weight-gradient shapes (GANTRON_PGEMM_DEBUG: gemm.hip)
for d in ${PROBE_SET:-0 8 1 2 4 5}; do echo "== GANTRON_PGEMM_DEBUG=$d"; GANTRON_PGEMM_DEBUG=$d PLANES=1 python tools/gbench.py 2>&1 | grep "planes\]"; done
