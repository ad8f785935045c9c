import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find decode kernels, take the window between the 2nd and 3rd persistent decode launch (one full bench step)
idx = [i for i, r in enumerate(rows) if 'decode_kernel<4, 0>' in r['Kernel_Name']]
print('decode launches', len(idx))
a, b = idx[1], idx[2]
t0 = int(rows[a]['End_Timestamp'])
busy = 0; prev_end = t0; gaps = []
acc = {}
for r in rows[a+1:b+1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gaps.append((s - prev_end, r['Kernel_Name'][:50]))
    n = r['Kernel_Name'][:50]
    acc[n] = acc.get(n, 0) + (e - s)
    prev_end = max(prev_end, e)
print('window us', (int(rows[b]['End_Timestamp']) - t0) / 1e3)
for n, v in sorted(acc.items(), key=lambda x: -x[1])[:14]: print('%9.1f us  %s' % (v / 1e3, n))
print('sum of gaps us', sum(g for g, _ in gaps) / 1e3)
for g, n in sorted(gaps, reverse=True)[:12]: print('gap %8.1f us before %s' % (g / 1e3, n))
