import csv, sys, collections, glob
f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    if pat and pat not in n: continue
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, d in agg.items():
    print(n, 'dispatches', len(next(iter(d.values()))))
    for c, v in sorted(d.items()):
        print(f'    {c:32s} avg {sum(v)/len(v):14.1f}')
