"""Aggregate rocprofv3 PMC passes into per-launch HBM traffic of the bench's kernel families.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d D -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d D -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py D/fetch_counter_collection.csv D/write_counter_collection.csv profiles/rNN_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  The two counters need separate passes (TCC slots)."""
import collections
import csv
import json
import sys

FAMILIES = {'attn_bwd_step_kernel': 'attn_bwd_step_kernel', 'plstm_fwd_kernel': 'plstm_fwd_kernel', 'plstm_bwd_kernel': 'plstm_bwd_kernel', 'attn_step2_fwd': 'attn_step2_fwd', 'xgemm_kernel': 'xgemm_kernel',
            'rec_cell_kernel': 'rec_cell_kernel', 'rec_cell_bf16_kernel': 'rec_cell_bf16_kernel', 'cell_bwd_query_kernel': 'cell_bwd_query_kernel', 'rec_gemm_kernel': 'rec_gemm_kernel', 'rec_gemm_h2_kernel': 'rec_gemm_h2_kernel', 'sgemm_kernel': 'sgemm_kernel',
            'attn_context_fwd_kernel': 'attn_context_fwd_kernel', 'attn_context_bwd_kernel': 'attn_context_bwd_kernel',
            'attn_energies_fwd': 'attn_energies_fwd', 'attn_energies_bwd': 'attn_energies_bwd',
            'attn_param_grads': 'attn_param_grads', 'rec_gemm_bf16_kernel': 'rec_gemm_bf16_kernel', 'hgemm_kernel': 'hgemm_kernel',
            'lstm_cell_fwd_kernel': 'lstm_cell_fwd_kernel', 'lstm_cell_bwd_kernel': 'lstm_cell_bwd_kernel'}


def aggregate(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            for key, fam in FAMILIES.items():
                if key in r['Kernel_Name']:
                    acc[fam][0] += 1
                    acc[fam][1] += float(r['Counter_Value'])
                    break
    return acc


def main(fetch_csv, write_csv, out_json):
    fetch, write = aggregate(fetch_csv, 'FETCH_SIZE'), aggregate(write_csv, 'WRITE_SIZE')
    out = {}
    for fam in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(fam, [0, 0.0])
        nw, vw = write.get(fam, [0, 0.0])
        fb = 2.0 * vf * 1024 / max(nf, 1)          # gfx950: FETCH_SIZE counts half of a wide coalesced stream
        wb = vw * 1024 / max(nw, 1)
        out[fam] = dict(launches=nf, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                        hbm_bytes_per_launch=fb + wb)
    json.dump(dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1',
                   correction='FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB -> bytes', kernels=out), open(out_json, 'w'), indent=1)
    for k, v in out.items():
        print(f"{k:28s} {v['launches']:6d} launches  fetch {v['fetch_bytes_per_launch'] / 1e6:8.2f} MB  write {v['write_bytes_per_launch'] / 1e6:7.2f} MB")


if __name__ == '__main__':
    main(*sys.argv[1:4])
