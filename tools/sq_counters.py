"""Per-kernel SQ counters of a `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES` run
(no trace domains in the same run): python tools/sq_counters.py <output directory>.  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles)."""
import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'])
    if (k, key) not in seen: seen.add((k, key)); cnt[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:8]:
    n = cnt[k]
    print(k, 'launches', n)
    for c, x in sorted(v.items()): print('   %-24s %14.0f per launch' % (c, x / n))
