#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the message-aggregate kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o r -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o r -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline
    python profiles/pmc_spmm.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/r01_pmc_spmm.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): both counters are in KB; FETCH_SIZE tallies the
128-byte requests of 16-byte-per-lane loads at 64 bytes and is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv, glob, json, sys


def avg(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'k_spmm' in r['Kernel_Name'] and r['Counter_Name'] == counter]
    return sum(v) / len(v), len(v)


fetch, n = avg(sys.argv[1], 'FETCH_SIZE')
write, _ = avg(sys.argv[2], 'WRITE_SIZE')
print(json.dumps({
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) -- python3 bench.py '
              '--steps 2 --warmup 3 --no-cpu-baseline --no-roofline',
    'kernel': 'k_spmm (all instantiations)', 'dispatches': n,
    'fetch_size_kb_raw_avg': round(fetch, 1), 'write_size_kb_avg': round(write, 1),
    'correction': 'FETCH_SIZE doubled (gfx950 counts the 128-B requests of 16-B-per-lane loads at 64 B, MI355X_MICROARCH.md HBM '
                  'section); WRITE_SIZE exact',
    'traffic_bytes_per_launch': int(round((2 * fetch + write) * 1024)),
    'note': 'per-launch average over the launch mix of 5 training steps (2 eager warm-up + 3 hipGraph replays)'}, indent=1))
