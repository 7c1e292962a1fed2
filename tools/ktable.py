"""Print the per-kernel table of a bench.py JSON line (file argument)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'full', d.get('full_search_step', {}).get('ms_per_pair'))
tot = 0.0
for k in d.get('roofline_kernels', []):
    print(f"{k['kernel']:24s} n={k['launches_per_step']:4.1f} avg={k['avg_us']:6.2f} "
          f"step={k['us_per_step']:6.2f} {k['bound']} frac={k['frac']:.3f}")
    tot += k['us_per_step']
print('sum of listed kernels (us):', round(tot, 1))
