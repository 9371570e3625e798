#!/usr/bin/env python3
"""Reads the two rocprofv3 --pmc passes over tools/micro/pmc_calib (FETCH_SIZE, WRITE_SIZE) and prints, per kernel, the counter in bytes
against the bytes the kernel is known to move: the factor tools/pmc_summary.py must apply for that access width."""
import csv, glob, os, sys
KNOWN = {'read8': ('FETCH_SIZE', 2 << 30), 'read16': ('FETCH_SIZE', 2 << 30), 'rowsum165': ('FETCH_SIZE', (2 << 30) // (168 * 8) * 165 * 8),
         'write8': ('WRITE_SIZE', 2 << 30), 'write16': ('WRITE_SIZE', 2 << 30)}


def read(dirname, counter):
    acc = {}
    for f in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') == counter:
                k = row['Kernel_Name'].split('(')[0]
                acc.setdefault((k, row['Dispatch_Id']), 0.)
                acc[(k, row['Dispatch_Id'])] += float(row['Counter_Value'])
    out = {}
    for (k, _), v in acc.items():
        out.setdefault(k, []).append(v)
    return dict((k, sum(v) / len(v)) for k, v in out.items())


def main():
    vals = {'FETCH_SIZE': read(sys.argv[1], 'FETCH_SIZE'), 'WRITE_SIZE': read(sys.argv[2], 'WRITE_SIZE')}
    for k, (counter, known) in KNOWN.items():
        got = [v for name, v in vals[counter].items() if k in name]
        if not got:
            print('%-10s no %s rows' % (k, counter)); continue
        kb = got[0]
        print('%-10s %-10s = %12.0f KiB = %6.3f GiB   known %6.3f GiB   known / counter = %.3f' % (k, counter, kb, kb / 2**20, known / 2**30, known / (kb * 1024.)))


if __name__ == '__main__':
    main()
