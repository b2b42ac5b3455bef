"""Averages of PMC counters per attention kernel from a rocprofv3 --pmc run of tools/bench_attn.py (diagnostics).
    python tools/pmc_attn.py <dir> [<dir> ...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'k_attn' in r['Kernel_Name']:
                acc[(r['Kernel_Name'].split('(')[0].split('::')[-1], r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            print(f'{k:28s} {c:24s} n={len(v):4d} avg={sum(v) / len(v):14.1f}')
