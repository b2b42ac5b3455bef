"""Per-kernel averages of PMC counters from rocprofv3 --pmc runs of tools/bench_attn.py, as JSON (profiles/r02_pmc_attn.json):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_attn_f -o r -- python3 tools/bench_attn.py 32 8 planes
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_attn_w ...;  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... -d gpurun_out/pmc_attn_h ...
    python tools/pmc_attn.py gpurun_out/pmc_attn_f gpurun_out/pmc_attn_w gpurun_out/pmc_attn_h
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes,
MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, sys, collections
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'k_attn' in n and 'edge_attrs' not in n:
                acc[(n[n.index('k_attn'):].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'k_attn' in n and 'edge_attrs' not in n:
                dur[n[n.index('k_attn'):].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = {'source': 'rocprofv3 --pmc <counter> --kernel-trace (separate passes) -- python3 tools/bench_attn.py 32 8 planes '
                 '(8 heads, C = 32, cfg4 mesh: N = 124464, E = 492606)', 'kernels': {}}
for k in sorted({k for k, _ in acc}):
    g = lambda c: (sum(acc[(k, c)]) / len(acc[(k, c)])) if acc.get((k, c)) else None
    f, w, h, m = g('FETCH_SIZE'), g('WRITE_SIZE'), g('TCC_HIT_sum'), g('TCC_MISS_sum')
    e = {'launches_sampled': len(acc.get((k, 'FETCH_SIZE'), [])), 'avg_us_under_pmc': round(sum(dur[k]) / len(dur[k]), 1) if dur.get(k) else None}
    if f is not None:
        e['fetch_bytes'] = int(2 * f * 1024)
    if w is not None:
        e['write_bytes'] = int(w * 1024)
    if h is not None and m is not None:
        e['l2_hit_rate'] = round(h / (h + m), 3)
    out['kernels'][k] = e
print(json.dumps(out, indent=1))
