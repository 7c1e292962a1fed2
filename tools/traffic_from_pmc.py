"""profiles/<tag>_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes — TCC has 4 slots and FETCH_SIZE costs 3).  Units: the counters are in KiB.  gfx950
correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half the bytes of a wide
coalesced 16-B-per-lane streaming read, so the read side is doubled for the float4-streaming
kernels; WRITE_SIZE is exact for 16-B streaming stores and float atomics.  Kernels whose loads are
dword-wide (the MFMA operand fetches) are uncalibrated: raw and doubled values are both kept."""
import collections
import csv
import json
import sys

fetch_csv, write_csv, out = sys.argv[1:4]
STREAMING = ('mixsum', 'cat_ln', 'ln_affine', 'node_mix', 'bn_', 'fold_weight', 'adam', 'backward_epilogue')
WRAPPER = [('mixsum_pair_fwd_k', 'mixsum_pair_fwd'), ('mixsum_pair_bwd_k', 'mixsum_pair_bwd'),
           ('node_mix_ln_fwd_k', 'node_mix_ln_fwd'), ('conv_fwd_sdpa_k', 'conv1x1_fwd_sdpa'), ('conv_pipe_fwd_sdpa_k', 'conv1x1_fwd_sdpa'),
           ('conv_bwd_all_pipe_k', 'conv1x1_bwd_all_sdpa'), ('conv_pipe_fwd_k', 'conv1x1_fwd'),
           ('conv_pipe_bwd_k', 'conv1x1_bwd_data'),
           ('conv_bwd_sdpa_k', 'conv1x1_bwd_data_sdpa'), ('conv_bwd_all_k', 'conv1x1_bwd_all_sdpa'),
           ('linear_bwd_k', 'linear_bwd'), ('backward_epilogue_k', 'backward_epilogue'), ('cell_prologue_k', 'cell_prologue'), ('adam_multi_k', 'adam_multi'),
           ('linear_fwd_k', 'linear_fwd'), ('mixsum_fwd_k', 'mixsum_fwd'), ('mixsum_bwd_k', 'mixsum_bwd'), ('cat_ln_fwd_k', 'cat_ln_fwd'),
           ('cat_ln_bwd_k', 'cat_ln_bwd'), ('ln_affine_bwd_k', 'ln_affine_bwd'), ('sdpa_ln_fwd_k', 'sdpa_ln_fwd'),
           ('sdpa_ln_bwd_k', 'sdpa_ln_bwd'), ('<true', 'conv1x1_fwd'), ('<false', 'conv1x1_bwd_data'),
           ('conv_w_k', 'conv1x1_bwd_weight'), ('node_mix_fwd_k', 'node_mix_fwd'), ('node_mix_bwd_k', 'node_mix_bwd'),
           ('bn_bwd_apply_k', 'bn_bwd_apply'), ('bn_finalize_k', 'bn_finalize'), ('fold_weight_k', 'fold_weight')]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


f, w = load(fetch_csv, 'FETCH_SIZE'), load(write_csv, 'WRITE_SIZE')
per_wrapper = collections.defaultdict(lambda: dict(launches=0, fetch_kib=0.0, write_kib=0.0))
for k in f:
    name = k.replace('(anonymous namespace)::', '')
    if 'bmnas' not in k and not any(p in name for p, _ in WRAPPER):
        continue
    wrap = next((wn for p, wn in WRAPPER if p in name), None)
    if wrap is None or (wrap.startswith('conv1x1_') and 'conv' not in name) or \
            (wrap in ('conv1x1_fwd', 'conv1x1_bwd_data') and 'ksplit' not in name and 'conv_nj' not in name
             and 'conv_lds' not in name and 'conv_pipe' not in name):
        continue
    d = per_wrapper[wrap]
    d['launches'] += len(f[k])
    d['fetch_kib'] += sum(f[k])
    d['write_kib'] += sum(w.get(k, [0.0]))
res = {}
for wrap, d in per_wrapper.items():
    n = d['launches']
    fetch, write = d['fetch_kib'] / n * 1024, d['write_kib'] / n * 1024
    streaming = any(wrap.startswith(s) for s in STREAMING)
    res[wrap] = {'fetch_raw_bytes': round(fetch), 'write_bytes': round(write),
                 'traffic_bytes': round((2 * fetch if streaming else fetch) + write),
                 'read_correction': 'x2 (16-B/lane streaming reads, gfx950)' if streaming else
                                    'none (dword MFMA-operand loads: uncalibrated, could be up to x2)',
                 'launches_sampled': n}
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(res.items()):
    print(f"{k:22s} traffic/launch {v['traffic_bytes']/1e6:8.2f} MB  (fetch raw {v['fetch_raw_bytes']/1e6:.2f} MB, write {v['write_bytes']/1e6:.2f} MB)")
