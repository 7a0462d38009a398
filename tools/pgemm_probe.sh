#!/bin/bash
# knock-outs of pgemm_kernel on the decoder weight-gradient shapes (GANTRON_DEBUG=pgemm=<bits>: gemm.hip)
for d in ${PROBE_SET:-0 8 1 2 4 5}; do echo "== GANTRON_DEBUG=pgemm=$d"; GANTRON_DEBUG=pgemm=$d PLANES=1 python tools/gbench.py 2>&1 | grep "planes\]"; done
