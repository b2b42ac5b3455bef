#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` output directory (csv, or the default rocpd sqlite database) into a short
markdown summary; kernels are grouped by full name (template instances stay separate).

    python profiles/summarize.py gpurun_out/prof_r1f 'command line' > profiles/r01_kernel_stats.md
"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r'\s+', ' ', name)
    m = re.search(r'(k_[a-z_0-9]+(?:<\d+>)?)', name)
    if m:
        return m.group(1)
    m = re.findall(r'(direct_copy_kernel|[A-Za-z_]*Functor[A-Za-z_]*|binary_internal::[A-Za-z]+|CatArray[A-Za-z_]*|reduce_kernel|'
                   r'multi_tensor_apply_kernel|rocclr_[A-Za-z]+|Cijk|distribution_[a-z_]+)', name)
    return 'torch: ' + ' '.join(dict.fromkeys(m)) if m else name[:60]


def db_rows(path):
    import sqlite3
    q = ('select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by name '
         'order by 3 desc')
    rows = sqlite3.connect(path).execute(q).fetchall()
    tot = sum(r[2] for r in rows)
    return [{'Name': n, 'Calls': c, 'TotalDurationNs': s, 'AverageNs': s / c, 'MinNs': mn, 'MaxNs': mx,
             'Percentage': 100.0 * s / tot} for n, c, s, mn, mx in rows]


def main(d, cmd):
    f = glob.glob(f'{d}/**/*_kernel_stats.csv', recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else db_rows(glob.glob(f'{d}/**/*_results.db', recursive=True)[0])
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print(f'# rocprofv3 kernel summary\n\n`{cmd}`\n\n{calls} kernel launches, {tot / 1e6:.1f} ms of kernel time.\n')
    print('| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|')
    for r in rows[:32]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.2f} | "
              f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
