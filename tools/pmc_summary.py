#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/traffic_rNN.json.

    python tools/pmc_summary.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run> <segments> <states> <restarts per launch> <out.json>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE counts wide coalesced
reads at half their size (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Kernel names are folded to the
names bench.py uses."""
import csv, glob, json, os, sys


def fold(name):
    for key, out in (('k_fbm', 'k_fb'), ('k_fbq', 'k_fb'), ('k_fbv', 'k_fb'), ('k_fbk', 'k_fb'), ('k_fb<', 'k_fb'), ('k_pairwise', 'k_pairwise')):
        if key in name:
            return out
    if 'k_cells' in name:
        # k_cells<NS, MODE, MASK, CACHE>: MODE 0 = frame log-probabilities, 1 = marginals, 2 = refresh,
        # 3 = marginals fused with the next sweep's frame pass (counted with the marginals, as in bench.py)
        args = name[name.index('<') + 1:name.index('>')].split(',')
        mode = int(args[1])
        return {0: 'k_framelogprob', 1: 'k_marginals<true>', 2: 'k_marginals<false>', 3: 'k_marginals<true>'}[mode]
    return None


def read(dirname, counter):
    acc = {}
    for f in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            k = fold(row['Kernel_Name'])
            if k is None:
                continue
            key = (k, row['Dispatch_Id'])
            acc[key] = acc.get(key, 0.) + float(row['Counter_Value'])
    out = {}
    for (k, _), v in acc.items():
        out.setdefault(k, []).append(v)
    return dict((k, sum(v) / len(v)) for k, v in out.items())


def main():
    fdir, wdir, seg, st, rst, outp = sys.argv[1:7]
    fetch, write = read(fdir, 'FETCH_SIZE'), read(wdir, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fetch) & set(write)):
        kernels[k] = {'fetch_size_kb': fetch[k], 'write_size_kb': write[k], 'hbm_bytes_per_launch': (2. * fetch[k] + write[k]) * 1024.}
    json.dump({'note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: '
                       'gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)',
               'workload': {'segments': int(seg), 'states': int(st), 'restarts': int(rst)}, 'kernels': kernels}, open(outp, 'w'), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == '__main__':
    main()
