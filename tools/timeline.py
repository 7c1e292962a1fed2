"""Timeline of one hipGraph replay from a rocprofv3 --kernel-trace CSV: per kernel start offset,
duration and the idle gap before it.  usage: timeline.py <kernel_trace.csv> [n_kernels_per_step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: walk back from the end until the first kernel name of the step repeats
names = [r['Kernel_Name'] for r in rows]
last = names[-1]
# find the period by looking for the previous occurrence pattern of the final 3 kernels
tail = names[-3:]
period = None
for p in range(10, 200):
    if names[-3 - p:-p] == tail and names[-3 - 2 * p:-2 * p] == tail:
        period = p
        break
if period is None:
    print('no period found')
    sys.exit(1)
step = rows[-period:]
prev_end = int(rows[-period - 1]['End_Timestamp'])
t0 = int(step[0]['Start_Timestamp'])
tot_k = tot_g = 0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = s - prev_end
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:46]
    print(f'{(s - t0) / 1e3:8.2f} us  dur {(e - s) / 1e3:6.2f}  gap {gap / 1e3:6.2f}  {n}')
    tot_k += e - s
    tot_g += max(gap, 0)
    prev_end = e
print(f'kernels {period}: busy {tot_k / 1e3:.1f} us, gaps {tot_g / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us')
