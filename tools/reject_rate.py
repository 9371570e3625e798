import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartGroups
R = 16
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, R, 8, num_clones=3)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    rs = RestartGroups(e, ps, 8, groups=1, num_clones=3, device=0, quiet=False, seeds=list(range(R)))
    el = rs.calculate_elbo()
    for m, v in zip(rs.models, el): m.prev_elbo = float(v)
    rs.run(5, 0, 5)
txt = buf.getvalue()
import collections
c = collections.Counter()
for line in txt.splitlines():
    if 'rejected' in line:
        c[line.split()[1] if line.split()[0][0].isdigit() else line.split()[0]] += 1
print('rejections over 5 EM iterations x 16 restarts:', dict(c))
print([ (m.prev_elbo) for m in rs.models][:4])
