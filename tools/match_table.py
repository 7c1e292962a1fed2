"""gpurun_out/match_step_outcomes.jsonl (written by tests/gpu_util.match_step during a -m gpu run) ->
tests/golden/match_step_table.json: label -> the largest number of flipped ReLU decisions any recorded run needed
(0 = matched the fp32 or float64 evaluation as it stands).  The tests cap later runs at that number + 1.
    python tools/match_table.py [outcomes.jsonl ...]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sys.argv[1:] or [os.path.join(ROOT, 'gpurun_out', 'match_step_outcomes.jsonl')]
table, kinds = {}, {}
for path in paths:
    for line in open(path):
        r = json.loads(line)
        if r['label'] in ('', 'self-test'):
            continue
        m = re.match(r'fp64\+(\d+)flips', r['outcome'])
        n = int(m.group(1)) if m else 0
        table[r['label']] = max(table.get(r['label'], 0), n)
        kinds.setdefault(r['label'], set()).add(r['outcome'])
# 'det: ...' labels (BMNAS_DETERMINISTIC runs, bit-reproducible): the KIND of evaluation matched, compared exactly
det = {k: sorted(v) for k, v in kinds.items() if k.startswith('det: ')}
bad = {k: v for k, v in det.items() if len(v) != 1}
if bad:
    raise SystemExit(f'deterministic labels with more than one recorded outcome (not reproducible?): {bad}')
table = {k: v for k, v in table.items() if not k.startswith('det: ')}
out = os.path.join(ROOT, 'tests', 'golden', 'match_step_table.json')
with open(out, 'w') as f:
    json.dump(dict(sorted(table.items())), f, indent=0)
print(f'{len(table)} labels -> {out}')
if det:
    out = os.path.join(ROOT, 'tests', 'golden', 'match_step_table_det.json')
    with open(out, 'w') as f:
        json.dump({k: v[0] for k, v in sorted(det.items())}, f, indent=0)
    print(f'{len(det)} deterministic labels -> {out}')
for k, v in sorted(kinds.items()):
    if v != {'fp32'}:
        print(f'  {k}: {sorted(v)}')
