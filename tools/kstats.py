import csv, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof*/runc/*_kernel_stats.csv'))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 35
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f, 'total us/step', tot / 1e3 / steps)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:58]:58s} n/step={float(r['Calls'])/steps:5.1f} avg={float(r['AverageNs'])/1e3:7.2f}us /step={float(r['TotalDurationNs'])/1e3/steps:7.2f}us {float(r['Percentage']):5.1f}%")
