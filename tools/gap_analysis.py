"""Idle time between the kernels of one captured training step, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -o g -- python3 bench.py --steps 20 --warmup 5 --repeats 0 --frozen-steps 0 --no-split-variant --no-cpu-baseline --no-roofline
    python tools/gap_analysis.py gpurun_out/gap
A step = the kernels between two k_flat_adam launches.  Prints busy time, span and the gap histogram of the median step."""
import csv, glob, sys
import numpy as np
files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'k_flat_adam' in r[2]]
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    ks = rows[a + 1:b + 1]
    if len(ks) < 100:
        continue
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = np.array([max(ks[i + 1][0] - ks[i][1], 0) for i in range(len(ks) - 1)])
    steps.append((span, busy, len(ks), gaps, ks))
steps.sort(key=lambda x: x[0])
span, busy, n, gaps, ks = steps[len(steps) // 2]
print(f'{len(steps)} steps; median step: {n} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us '
      f'({100 * (span - busy) / span:.1f} %)')
print('gap histogram (us): ' + ', '.join(f'<{hi}: {int(((gaps >= lo * 1e3) & (gaps < hi * 1e3)).sum())}' for lo, hi in [(0, 0.5), (0.5, 1), (1, 2), (2, 4), (4, 8), (8, 1e9)]))
big = np.argsort(-gaps)[:12]
for i in big:
    print(f'   gap {gaps[i] / 1e3:6.2f} us after {ks[i][2][:60]:60s} before {ks[i + 1][2][:50]}')
# gap by predecessor kernel family
fam = {}
for i, g in enumerate(gaps):
    k = ks[i][2].split('(')[0].split('<')[0][-40:]
    fam.setdefault(k, [0, 0.0]); fam[k][0] += 1; fam[k][1] += g
for k, (c, g) in sorted(fam.items(), key=lambda x: -x[1][1])[:12]:
    print(f'   after {k:42s} {c:4d} gaps, {g / 1e3:7.1f} us total, {g / c / 1e3:5.2f} us each')
if len(sys.argv) > 2:          # the median step's kernel sequence: index, duration (us), short name
    import re
    with open(sys.argv[2], 'w') as fh:
        for i, (s, e, k) in enumerate(ks):
            k = re.sub(r'\(anonymous namespace\)::', '', k)
            k = re.sub(r'^void ', '', k)
            fh.write(f'{i:4d} {(e - s) / 1e3:7.2f} {k[:110]}\n')
