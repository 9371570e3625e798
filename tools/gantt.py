"""Text Gantt chart of a rocprofv3 kernel trace (`*_kernel_trace.csv`): per hardware queue, the kernels of a window in the middle of
the run in start order -- long kernels one line each, runs of short kernels (< SHORT us) folded into one line -- so that one EM period of
the restart groups can be read: what runs next to what, who waits for whom.  Usage: python tools/gantt.py trace.csv [window_ms] [short_us]"""
import csv
import collections
import sys


def short(name):
    name = name.split('(')[0].replace('void ', '')
    return name[:34]


def main():
    path = sys.argv[1]
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 80.
    SHORT = float(sys.argv[3]) if len(sys.argv) > 3 else 150.
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']))
    rows.sort()
    t0 = rows[len(rows) // 2][0]
    t1 = t0 + int(win * 1e6)
    inside = [r for r in rows if r[0] >= t0 and r[0] < t1]
    queues = sorted(set(r[3] for r in inside), key=lambda q: -sum(1 for r in inside if r[3] == q))
    col = dict((q, i) for i, q in enumerate(queues))
    print('window %.1f ms from the middle of the trace; queues (columns): %s' % (win, ' '.join(queues)))
    print('%9s %9s  %s' % ('start ms', 'dur us', 'kernel (column = hardware queue)'))
    pend = {}   # queue -> [start, end, count, names]

    def flush(q):
        p = pend.pop(q, None)
        if p:
            names = collections.Counter(p[3]).most_common(3)
            print('%9.3f %9.0f  %s[%d short: %s]' % ((p[0] - t0) / 1e6, (p[1] - p[0]) / 1e3, '    ' * col[q] + ' ' * 36 * col[q], p[2], ', '.join('%s x%d' % nc for nc in names)))
    for s, e, name, q in inside:
        dur = (e - s) / 1e3
        if dur < SHORT:
            p = pend.get(q)
            if p is None:
                pend[q] = [s, e, 1, [short(name)]]
            else:
                p[1] = max(p[1], e); p[2] += 1; p[3].append(short(name))
            continue
        flush(q)
        print('%9.3f %9.0f  %s%s' % ((s - t0) / 1e6, dur, '    ' * col[q] + ' ' * 36 * col[q], short(name)))
    for q in list(pend):
        flush(q)


if __name__ == '__main__':
    main()
