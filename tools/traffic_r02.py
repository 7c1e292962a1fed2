"""profiles/<tag>_traffic.json, keyed by kernel (short name as in bench.py's roofline_kernels), from two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes: TCC has 4 slots, FETCH_SIZE costs 3).
Counter units are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly
half the bytes of a wide coalesced 16-B-per-lane streaming read, so the read side is doubled for the
float4-streaming kernels; WRITE_SIZE is exact for 16-B streaming stores and float atomics.  Kernels
whose loads are mostly dword-wide (MFMA operand fetches) are uncalibrated: raw value kept, noted.
usage: traffic_r02.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys

fetch_csv, write_csv, out = sys.argv[1:4]
STREAMING = ('mixsum', 'cat_ln', 'ln_affine', 'node_mix', 'bn_', 'fold_weight', 'adam', 'backward_epilogue',
             'cell_prologue', 'sum_chunks')


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    i = n.find('(')
    return n if i < 0 else n[:i]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return agg


f, w = load(fetch_csv, 'FETCH_SIZE'), load(write_csv, 'WRITE_SIZE')
res = {}
for k in sorted(f):
    if not (k.endswith('_k') or '_k<' in k):
        continue                                     # ours only (aten / runtime kernels left out)
    n = len(f[k])
    fetch = sum(f[k]) / n * 1024
    write = sum(w.get(k, [0.0])) / max(len(w.get(k, [0.0])), 1) * 1024
    streaming = any(k.startswith(s) for s in STREAMING)
    res[k] = {'fetch_raw_bytes': round(fetch), 'write_bytes': round(write),
              'traffic_bytes': round((2 * fetch if streaming else fetch) + write),
              'read_correction': 'x2 (16-B/lane streaming reads, gfx950)' if streaming else
                                 'none (dword / mixed-width operand loads: uncalibrated, could be up to x2)',
              'launches_sampled': n}
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print(f"{k:40s} traffic/launch {v['traffic_bytes'] / 1e6:8.2f} MB  (fetch raw {v['fetch_raw_bytes'] / 1e6:.2f} MB, "
          f"write {v['write_bytes'] / 1e6:.2f} MB)")
