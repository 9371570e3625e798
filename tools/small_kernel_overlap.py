"""What slows the M-step's small kernels next to the other group's sweeps: from a rocprofv3 kernel trace of the bench, the durations of one small kernel
(default k_ell_search_multi) split by the large kernel that was running when it started.  Usage: python tools/small_kernel_overlap.py trace.csv [name]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
key = sys.argv[2] if len(sys.argv) > 2 else 'k_ell_search_multi'
big = [r for r in rows if r[1] - r[0] > 300e3]
small = [r for r in rows if key in r[2]]
small = small[len(small) // 5:]
by = collections.defaultdict(list)
for s0, s1, _ in small:
    conc = sorted(set(b[2][:24] for b in big if b[0] <= s0 < b[1]))
    by[' + '.join(conc) if conc else '(nothing large)'].append((s1 - s0) / 1e3)
print('%s: %d launches' % (key, len(small)))
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    print('  %-70s n=%4d  median %7.1f us  mean %7.1f  p90 %7.1f' % (k, len(v), v[len(v) // 2], sum(v) / len(v), v[int(len(v) * 0.9)]))
