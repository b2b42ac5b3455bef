"""Average of one PMC counter per (kernel, grid) from a rocprofv3 --pmc run (counter_collection csv)."""
import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
pat = sys.argv[2]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        k = (r['Kernel_Name'].split('(')[0].split('::')[-1][:30], r['Grid_Size'], r['Counter_Name'])
        d.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in d.items():
    print(k, len(v), 'avg', round(sum(v) / len(v), 1))
