"""Durations of one kernel and the gaps between its consecutive launches in a rocprofv3 kernel trace: python tools/kernel_gaps.py <..._kernel_trace.csv> <kernel name substring>
(how long is a round of the device-driven parameter searches, and how close do back-to-back kernels of a stream start?)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2]
sel = [r for r in rows if name in r['Kernel_Name']]
sel.sort(key=lambda r: int(r['Start_Timestamp']))
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel]
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(sel, sel[1:])]
gaps = [g for g in gaps if g < 200000]
import statistics as st
print(name, 'n', len(sel), 'dur mean %.1f us median %.1f p90 %.1f' % (st.mean(d) / 1e3, st.median(d) / 1e3, sorted(d)[int(len(d) * .9)] / 1e3))
print('gap to next (same kernel, < 200 us): n', len(gaps), 'mean %.1f us median %.1f' % (st.mean(gaps) / 1e3, st.median(gaps) / 1e3))
# by grid size
by = collections.defaultdict(list)
for r, x in zip(sel, d):
    by[(r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))].append(x)
for k, v in sorted(by.items()):
    print('  grid', k, 'n', len(v), 'mean %.1f us' % (st.mean(v) / 1e3))
