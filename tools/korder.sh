#!/bin/bash
# kernel order of one replayed step:  tools/korder.sh [bench args]   (rocprofv3 kernel trace of bench.py; prints the
# last replay's dispatches in order with their durations)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/korder && rocprofv3 --kernel-trace --output-format csv -d /tmp/korder -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --regions 1 --no-full-step --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/korder/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
marks = [i for i, r in enumerate(rows) if 'cell_prologue_pair_k' in r[2] or r[2].startswith('cell_prologue_k') or 'cell_prologue_k(' in r[2]]
pm = [i for i, r in enumerate(rows) if 'cell_prologue_pair_k' in r[2]]
per = pm[-1] - pm[-2]
last = rows[-per:]
t0 = last[0][0]
for s, e, n in last:
    print(f'{(s - t0) / 1e3:8.2f} us  +{(e - s) / 1e3:6.2f}  {n[:90]}')
print('span', (last[-1][1] - t0) / 1e3)
PY
