"""Print the per-kernel table of a bench.py JSON line (file argument)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'full', d.get('full_search_step', {}).get('ms_per_pair'))
print('check', d.get('roofline_check'), d.get('roofline_error'))
tot = 0.0
for k in d.get('roofline_kernels', []):
    frac = '  -  ' if k['frac'] is None else f"{k['frac']:.3f}"
    print(f"{k['kernel'][:40]:40s} n={k['launches_per_step']:4.1f} avg={k['avg_us']:6.2f} "
          f"step={k['us_per_step']:6.2f} {k['bound']:7s} frac={frac}")
    tot += k['us_per_step']
print('sum of listed kernels (us):', round(tot, 1))
