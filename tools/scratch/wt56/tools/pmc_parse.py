import csv, glob, sys, collections
d = sys.argv[1]; frag = sys.argv[2]
f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if frag in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}')
