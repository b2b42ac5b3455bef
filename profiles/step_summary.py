#!/usr/bin/env python3
"""Whole-step summary of the captured training step from a rocprofv3 kernel trace plus the PMC traffic record of the same build:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_prof -o r -- python3 bench.py --steps 20 --warmup 5 ...
    python profiles/step_summary.py gpurun_out/r5_prof profiles/r05_pmc_traffic.json > profiles/r05_step_summary.json

A step = the kernels between two k_flat_adam launches (one hipGraph replay); the MEDIAN step by span is reported: launches,
launches shorter than 10 us (the launch-floor tail), busy / span time, and -- from the PMC record (FETCH_SIZE doubled + WRITE_SIZE
per launch of every kernel, profiles/pmc_traffic.py) -- the HBM-side bytes of one step, the rate they imply and the time the same
bytes would take at the 6.3 TB/s a plain copy reaches on this chip (MI355X_MICROARCH.md).  bench.py attaches the record as `step`.
"""
import csv
import glob
import json
import re
import sys

COPY_RATE_TBS = 6.3


def key(full):
    m = re.search(r'(k_\w+(?:<[^>]*>)?)', full)
    return m.group(1) if m else None


def main(trace_dir, pmc_json):
    rows = []
    for f in glob.glob(trace_dir + '/**/*kernel_trace.csv', recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if 'k_flat_adam' in r[2]]
    steps = []
    for a, b in zip(ends[:-1], ends[1:]):
        ks = rows[a + 1:b + 1]
        if len(ks) >= 100:
            steps.append((ks[-1][1] - ks[0][0], ks))
    steps.sort(key=lambda x: x[0])
    span, ks = steps[len(steps) // 2]
    durs = [e - s for s, e, _ in ks]
    small = [d for d in durs if d < 10000]
    pmc = json.load(open(pmc_json))['kernels']
    per_name, unpriced = {}, 0
    for (s, e, n) in ks:
        k = key(n)
        if k in pmc:
            per_name[k] = per_name.get(k, 0) + 1
        else:
            unpriced += 1
    step_bytes = sum(pmc[k]['traffic_bytes_per_launch'] * c for k, c in per_name.items())
    out = {
        'source': f'median of {len(steps)} replayed steps of a rocprofv3 --kernel-trace run of bench.py; bytes: {pmc_json} (per-launch '
                  'FETCH_SIZE x 2 + WRITE_SIZE of every k_* kernel x its launches in the step)',
        'launches': len(ks), 'launches_lt_10us': len(small), 'launches_lt_10us_ms': round(sum(small) / 1e6, 3),
        'kernel_busy_ms': round(sum(durs) / 1e6, 3), 'span_ms_under_profiler': round(span / 1e6, 3),
        'launches_without_pmc_record': unpriced,
        'hbm_bytes': int(step_bytes), 'hbm_gb': round(step_bytes / 1e9, 2),
        'implied_tbs_under_profiler': round(step_bytes / (span * 1e-9) / 1e12, 2),
        'floor_ms_at_copy_rate': round(step_bytes / (COPY_RATE_TBS * 1e12) * 1e3, 3), 'copy_rate_tbs': COPY_RATE_TBS,
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
