"""Kernel sequence of ONE hipGraph replay of the training step from a rocprofv3 --kernel-trace csv: the last complete
stretch of kernels between two FusedAdam launches.  Prints name (short), duration and the gap to the previous kernel, and a
per-name summary (diagnostics: finds the torch glue kernels and what surrounds them)."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\s+', ' ', n)
    m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', n)
    if m: return m.group(1)
    m = re.search(r'(multi_tensor_apply_kernel|CatArrayBatchedCopy[a-z_]*|index_elementwise_kernel|direct_copy_kernel|FillFunctor|CUDAFunctor_add|rocclr_[A-Za-z]+|reduce_kernel|[A-Za-z_]+Functor[A-Za-z_]*|distribution_[a-z_]+)', n)
    return 'torch:' + (m.group(1) if m else n[:50])
adam = [i for i, r in enumerate(rows) if 'FusedAdam' in r['Kernel_Name'] or 'k_flat_adam' in r['Kernel_Name']]
# steps are delimited by the last Adam kernel of a step: group consecutive adam launches
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 5]
a, b = ends[-2] + 1, ends[-1] + 1
seq = rows[a:b]
t0 = int(seq[0]['Start_Timestamp'])
print(f'{len(seq)} kernels, span {(int(seq[-1]["End_Timestamp"]) - t0) / 1e3:.1f} us')
tot = collections.OrderedDict()
prev_end = t0
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = short(r['Kernel_Name'])
    if len(sys.argv) > 2:
        print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.2f} gap {(s - prev_end) / 1e3:6.2f}  {nm}  grid {r["Grid_Size_X"]}')
    prev_end = e
    d = tot.setdefault(nm, [0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3
print('--- per kernel name')
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:60s} {n:4d} {us:9.1f} us  avg {us / n:7.2f}')
print('total kernel us', round(sum(v[1] for v in tot.values()), 1))
