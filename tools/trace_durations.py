"""Average device-side duration per (kernel, grid) from a rocprofv3 --kernel-trace csv (diagnostics)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
d = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if pat in n:
        k = (n.split('::')[-1][:40], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v2 = sorted(v)
    print(k, len(v), 'avg', round(sum(v) / len(v), 2), 'med', round(v2[len(v2) // 2], 2), 'min', round(v2[0], 2))
