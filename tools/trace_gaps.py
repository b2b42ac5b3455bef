"""Busy / idle breakdown of the steady-state part of a rocprofv3 kernel trace (diagnostics)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
# steady state: last 40 % of the trace
n = len(ev); ev = ev[int(n * 0.6):]
span = ev[-1][1] - ev[0][0]
busy = 0; cur_end = ev[0][0]; gaps = []; gapby = collections.Counter(); cnt = collections.Counter()
for i, (s, e, name) in enumerate(ev):
    if s > cur_end:
        gaps.append(s - cur_end)
        short = name.split('(')[0].split('::')[-1][:40]
        gapby[short] += s - cur_end; cnt[short] += 1
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print(f'kernels {len(ev)} span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms idle {(span-busy)/1e6:.2f} ms ({100*(span-busy)/span:.1f} %)')
gaps.sort()
print('gap median', gaps[len(gaps)//2]/1e3, 'us  mean', sum(gaps)/len(gaps)/1e3, 'us  p90', gaps[int(len(gaps)*.9)]/1e3, 'n', len(gaps))
print('idle before kernel (top):')
for k, v in gapby.most_common(12):
    print(f'  {v/1e3:9.1f} us total, {v/cnt[k]/1e3:6.2f} us avg x {cnt[k]:5d}  {k}')
