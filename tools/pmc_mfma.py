"""Matrix-pipe utilisation per kernel family from one rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d D -o mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_mfma.py D/mfma_counter_collection.csv profiles/rNN_pmc_mfma.json

Formula.  SQ_VALU_MFMA_BUSY_CYCLES is the number of shader cycles a SIMD's matrix pipe was busy, summed over every SIMD
the dispatch ran on (MI355X_MICROARCH.md, cycle-constants table: "= 32 x N_mfma for 32x32x16 bf16"; for the fp32 forms
64 x N for 32x32x2 and 32 x N for 16x16x4).  A dispatch of duration t (End - Start timestamp of the SAME row, ns) on the
whole chip has 256 CUs x 4 SIMDs x t x f_clk pipe-cycles available, so

    mfma_util = busy_cycles / (1024 x t x f_clk),        f_clk = 2.4 GHz (the part's maximum clock).

The clock is the stated constant, not a measurement: under load the chip runs below it (DVFS), so the figure is a LOWER
bound of the pipe's busy fraction in real cycles.  GRBM_GUI_ACTIVE is deliberately not used as the denominator: the guide
warns that it reads high on dispatches shorter than ~0.3 ms, which every loop kernel is.  Counter passes serialise the
kernels, so `avg_us` is the duration of an uncontended launch."""
import collections
import csv
import json
import sys

F_CLK = 2.4e9
SIMDS = 256 * 4
FAMILIES = ('attn_bwd_step_kernel', 'plstm_fwd_kernel', 'plstm_bwd_kernel', 'attn_step2_fwd', 'xgemm_kernel', 'rec_cell_kernel', 'rec_cell_bf16_kernel', 'rec_gemm_kernel', 'rec_gemm_h2_kernel', 'rec_gemm_bf16_kernel', 'sgemm_kernel', 'hgemm_kernel', 'attn_energies_fwd',
            'attn_energies_bwd', 'attn_context_fwd_kernel', 'attn_context_bwd_kernel', 'attn_param_grads', 'cell_bwd_query_kernel',
            'cell_query_kernel', 'lstm_cell_fwd_kernel', 'lstm_cell_bwd_kernel')


def main(counter_csv, out_json):
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    with open(counter_csv) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != 'SQ_VALU_MFMA_BUSY_CYCLES':
                continue
            fam = next((k for k in FAMILIES if k in r['Kernel_Name']), None)
            if fam is None:
                continue
            a = acc[fam]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
            a[2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    out = {}
    for fam, (n, busy, secs) in sorted(acc.items()):
        out[fam] = dict(launches=n, mfma_busy_cycles_per_launch=busy / n, avg_us=secs / n * 1e6,
                        mfma_util=busy / (SIMDS * secs * F_CLK) if secs > 0 else 0.0)
        print(f'{fam:26s} {n:7d} launches  {secs / n * 1e6:9.2f} us  mfma_util {out[fam]["mfma_util"]:.3f}')
    json.dump(dict(source='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace of bench.py --steps 1 --warmup 1 --no-cpu-baseline',
                   formula='mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch duration x 2.4e9 Hz); duration = End - Start '
                           'timestamp of the dispatch; the clock is the nominal maximum (a lower bound of the busy fraction under DVFS)',
                   f_clk_hz=F_CLK, simds=SIMDS, kernels=out), open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:3])
