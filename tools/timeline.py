"""Concurrency picture of a bench run from a rocprofv3 kernel trace (`*_kernel_trace.csv`): over the window that holds
the middle half of the forward-backward launches, which share of the wall time has 0 / 1 / 2 forward-backward kernels in
flight, how much has no kernel at all, and the busy time per kernel.  Usage: python tools/timeline.py trace.csv [fb-name-substring]"""
import csv
import collections
import sys


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else 'k_fbm<'
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']))
    fb = sorted(r for r in rows if key in r[2])
    n = len(fb)
    lo, hi = fb[n // 4][0], fb[3 * n // 4][0]
    inside = [r for r in rows if r[1] > lo and r[0] < hi]
    ev = []
    for s, e, name, q in inside:
        s, e = max(s, lo), min(e, hi)
        isfb = key in name
        ev.append((s, 1, isfb)); ev.append((e, -1, isfb))
    ev.sort()
    any_n = fb_n = 0
    last = lo
    by_fb = collections.Counter(); by_any = collections.Counter(); other_while_fb = 0
    for t, d, isfb in ev:
        dt = t - last
        by_fb[fb_n] += dt; by_any[min(any_n, 4)] += dt
        if fb_n and any_n > fb_n:
            other_while_fb += dt
        last = t
        any_n += d
        if isfb:
            fb_n += d
    wall = hi - lo
    print('window %.1f ms, %d forward-backward launches in it' % (wall / 1e6, n // 2))
    print('forward-backward kernels in flight: ' + ', '.join('%d: %.1f%%' % (k, 100. * v / wall) for k, v in sorted(by_fb.items())))
    print('kernels of any kind in flight:      ' + ', '.join('%s%d: %.1f%%' % ('>=' if k == 4 else '', k, 100. * v / wall) for k, v in sorted(by_any.items())))
    print('another kernel next to a forward-backward kernel: %.1f%% of the window' % (100. * other_while_fb / wall))
    busy = collections.Counter(); cnt = collections.Counter()
    for s, e, name, q in inside:
        busy[name.split('(')[0][:60]] += min(e, hi) - max(s, lo); cnt[name.split('(')[0][:60]] += 1
    for name, v in busy.most_common(14):
        print('  %-62s %6.1f%% of wall  n=%d avg %.3f ms' % (name, 100. * v / wall, cnt[name], v / cnt[name] / 1e6))


if __name__ == '__main__':
    main()
