#!/usr/bin/env python3
"""Per-launch HBM-side traffic of EVERY kernel of the training step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o r -- python3 bench.py --steps 2 --warmup 3 --frozen-steps 0 --repeats 0 --no-split-variant --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o r -- python3 bench.py (same flags)
    python profiles/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/r03_pmc_traffic.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): both counters are in KB; FETCH_SIZE tallies the
128-byte requests of 16-byte-per-lane loads at 64 bytes and is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
`traffic_bytes_per_launch` (top level) is the launch-weighted mean over the clip-resident message-aggregate launches
(k_cheb_clip), the kernel bench.py's `roofline` object prices.
"""
import collections, csv, glob, json, re, sys


def key(full):
    m = re.search(r'(k_\w+(?:<[^>]*>)?)', full)
    return m.group(1) if m else None


def avg(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'])
        if k and r['Counter_Name'] == counter:
            acc[k].append(float(r['Counter_Value']))
    return acc


fetch, write = avg(sys.argv[1], 'FETCH_SIZE'), avg(sys.argv[2], 'WRITE_SIZE')
cmd = sys.argv[3] if len(sys.argv) > 3 else ('python3 bench.py --steps 2 --warmup 3 --frozen-steps 0 --repeats 0 --no-split-variant '
                                            '--no-cpu-baseline --no-roofline')
kernels = {}
for k in sorted(fetch, key=lambda k: -sum(fetch[k]) - sum(write.get(k, [0]))):
    f, w = fetch[k], write.get(k, [0.0])
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    kernels[k] = {'dispatches': len(f), 'fetch_size_kb_raw_avg': round(fa, 1), 'write_size_kb_avg': round(wa, 1),
                  'traffic_bytes_per_launch': int(round((2 * fa + wa) * 1024))}
clip = [k for k in kernels if k.startswith('k_cheb_clip')]
n = sum(kernels[k]['dispatches'] for k in clip)
top = int(round(sum(kernels[k]['traffic_bytes_per_launch'] * kernels[k]['dispatches'] for k in clip) / n)) if n else None
print(json.dumps({
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- ' + cmd,
    'correction': 'FETCH_SIZE doubled (gfx950 counts the 128-B requests of 16-B-per-lane loads at 64 B, MI355X_MICROARCH.md HBM '
                  'section); WRITE_SIZE exact',
    'kernel': 'k_cheb_clip (forward and backward instantiations, launch-weighted)', 'traffic_bytes_per_launch': top,
    'note': 'per-launch averages over the launch mix of 5 training steps (2 eager warm-up + 3 hipGraph replays)',
    'kernels': kernels}, indent=1))
