"""Average PMC counter values per kernel from a rocprofv3 --pmc run (csv output): python tools/pmc_kernels.py <dir> [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if sub in k:
        acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'    {c:34s} n={len(v):4d} avg={sum(v) / len(v):14.1f}')
