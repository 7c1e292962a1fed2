"""profiles/<tag>_traffic.json, keyed by kernel (short name as in bench.py's roofline_kernels), from two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes: TCC has 4 slots, FETCH_SIZE costs 3).
Counter units are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the L2's
128-byte fabric requests at 64 bytes, so the read side is doubled — for EVERY kernel: the calibration in
profiles/r02_fetch_calibration.txt (tools/calibrate_fetch.sh: a 256 MiB buffer read once with 4 / 8 / 16
bytes per lane and as the GEMM kernels' 64-byte operand rows) gives x2.000 for all dense patterns, and
shows that rows using 64 of every 128 bytes still move whole 128-byte lines.  WRITE_SIZE is exact for
16-B streaming stores and float atomics.
usage: traffic_two_pass.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys

fetch_csv, write_csv, out = sys.argv[1:4]


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
    res[k] = {'fetch_raw_bytes': round(fetch), 'write_bytes': round(write),
              'traffic_bytes': round(2 * fetch + write),
              'read_correction': 'x2 (128-B fabric requests tallied at 64 B; profiles/r02_fetch_calibration.txt)',
              'launches_sampled': n}
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print(f"{k:40s} traffic/launch {v['traffic_bytes'] / 1e6:8.2f} MB  (fetch raw {v['fetch_raw_bytes'] / 1e6:.2f} MB, "
          f"write {v['write_bytes'] / 1e6:.2f} MB)")
