"""Per-wave cycle stamps of the forward lattice k_viterbi_max (a library built with -DRMX_VIT_STAMPS; VERDICT r4 item 3: instrument, then halve):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRMX_VIT_STAMPS -o tools/micro/lib_vitstamps.so remixt_amd/csrc/rmx_api.hip
    python tools/vit_stamps.py [MAXCN]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remixt_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', os.environ.get('VITLIB', 'lib_vitstamps_base.so'))
from remixt_amd import bpmodel, synthetic
from remixt_amd.restarts import RestartSet
bpmodel.set_default_option('fb_debug', 1)
mcn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
import time
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=mcn, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 4, mcn)
rs = RestartSet(e, ps, mcn, num_clones=3, quiet=True, seeds=[0, 1, 2, 3])
b = rs.batch
b.variational_update(2)
b.infer_cn_batch(0, 4)
t0 = time.perf_counter(); b.infer_cn_batch(0, 4); print('decode of 4 restarts: %.1f ms wall (%s)' % ((time.perf_counter() - t0) * 1e3, os.path.basename(_lib.LIB_PATH)))
steps = b.info(24)
names = ('wait for the ring slot', 'LDS reads + add / max', 'merge of the partial maxima', 'row write + store', 'barrier')
print('k_viterbi_max, lattice kernel %d, %d steps; shader cycles per step (s_memtime)' % (b.info(14), steps))
for slot, wname in enumerate(('wave 0', 'middle wave', 'last wave')):
    v = [b.info(28 + slot * 5 + i) for i in range(5)]
    tot = sum(v)
    print('  %-12s ' % wname + '  '.join('%s %.0f' % (n, x / float(max(steps, 1))) for n, x in zip(names, v)) + '   | total %.0f cycles per step' % (tot / float(max(steps, 1))))
