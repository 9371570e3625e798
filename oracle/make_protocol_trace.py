"""Writes tests/golden/protocol_trace_*.npz: the CALL TRACE of the reference's own host class on its kernel object.

TEST INFRASTRUCTURE, build container only (imports the reference in place through oracle/refload.py; nothing of the
reference's source or bytecode is written anywhere -- the fixture is data: names, argument arrays, returned values).

INTEGRATION.md option A puts `remixt_amd.bpmodel.RemixtModel` under the reference's unmodified `BreakpointModel`
(remixt/cn_model.py:368-404, 444-569).  That pairing can run nowhere: the reference never reaches a GPU box and the HIP
library has no CPU form.  What CAN be recorded here is everything the reference's `BreakpointModel.fit` / `optimal_cn` /
`get_model_data` do to their kernel object: a recording proxy stands where `remixt.bpmodel.RemixtModel` stood while the
reference's own `fit` runs, and every constructor call, method call (arguments, return value, arrays written into
caller-provided outputs, exception type), attribute read (value) and attribute write (value) goes into the fixture in order.
tests/test_hip_protocol_trace.py replays the trace call by call against `remixt_amd.bpmodel.RemixtModel` on the GPU.

  python oracle/make_protocol_trace.py            # all cases
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import build_ref, refload          # noqa: E402
from oracle.make_golden import quiet           # noqa: E402
from remixt_amd import synthetic               # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


class Recorder(object):
    """Events in order + arrays stored once per content."""

    def __init__(self):
        self.events = []
        self.arrays = {}
        self._by_hash = {}

    def ref(self, v):
        """A JSON-able reference to a value: scalars inline, arrays by key (deduplicated by content)."""
        if v is None:
            return None
        if isinstance(v, (bool, np.bool_)):
            return {'v': bool(v), 't': 'bool'}
        if isinstance(v, (int, np.integer)):
            return {'v': int(v), 't': 'int'}
        if isinstance(v, (float, np.floating)):
            return {'v': repr(float(v)), 't': 'float'}
        if isinstance(v, str):
            return {'v': v, 't': 'str'}
        a = np.array(np.asarray(v))          # memoryview slices, ndarray views: a private copy
        h = hashlib.sha1(a.tobytes() + str((a.dtype, a.shape)).encode()).hexdigest()
        key = self._by_hash.get(h)
        if key is None:
            key = 'a%d' % len(self.arrays)
            self._by_hash[h] = key
            self.arrays[key] = a
        return {'a': key}

    def add(self, **ev):
        self.events.append(ev)


def make_proxy_class(real_cls, rec):
    class Proxy(object):
        def __init__(self, *args):
            rec.add(op='construct', args=[rec.ref(a) for a in args])
            object.__setattr__(self, '_m', real_cls(*args))

        def __getattr__(self, name):
            m = object.__getattribute__(self, '_m')
            v = getattr(m, name)
            if callable(v):
                def call(*args):
                    before = [rec.ref(a) for a in args]
                    ev = dict(op='call', name=name, args=before)
                    try:
                        ret = v(*args)
                    except Exception as err:          # the reference's ValueError / AssertionError protocol
                        ev['raises'] = type(err).__name__
                        rec.add(**ev)
                        raise
                    ev['ret'] = rec.ref(ret)
                    # caller-provided outputs: what the arrays hold after the call, where it differs from before
                    after = [rec.ref(a) if isinstance(a, np.ndarray) else None for a in args]
                    ev['out'] = dict((str(i), after[i]) for i in range(len(args)) if after[i] is not None and after[i] != before[i])
                    rec.add(**ev)
                    return ret
                rec.add(op='getmethod', name=name)
                return call
            rec.add(op='get', name=name, value=rec.ref(v), kind=type(v).__name__)
            return v

        def __setattr__(self, name, value):
            rec.add(op='set', name=name, value=rec.ref(value))
            setattr(object.__getattribute__(self, '_m'), name, value)

        def __dir__(self):
            names = dir(object.__getattribute__(self, '_m'))
            rec.add(op='dir', names=[n for n in names if not n.startswith('__')])
            return names
    return Proxy


def trace_case(name, N, M, max_cn, chains, seed, normal_contamination=True, zero_alleles=(), fit_seed=7, num_em_iter=2, num_update_iter=2):
    bp = refload.load_ref_bpmodel()
    cm = refload.load_ref_cn_model()
    e = synthetic.make_experiment(N, num_clones=M, max_copy_number=max_cn, num_chains=chains, seed=seed)
    x = e.x.copy()
    for n in zero_alleles:
        x[n, 0:2] = 0.
    p = synthetic.make_init_params(e, 1, max_cn, num_clones=M)[0]
    h_init = synthetic.h_init_from_params(p, M)
    rec = Recorder()
    real = bp.RemixtModel
    bp.RemixtModel = make_proxy_class(real, rec)
    try:
        with quiet():
            m = cm.BreakpointModel(x, e.l, e.adjacencies, e.breakpoints, max_copy_number=max_cn, divergence_weight=p['divergence_weight'],
                                   max_depth=p['max_depth'], normal_contamination=normal_contamination)
            m.num_em_iter = num_em_iter; m.num_update_iter = num_update_iter
            np.random.seed(fit_seed)
            m.fit(h_init)                  # remixt/cn_model.py:354-428
            m.optimal_cn()                 # :571-604 (infer_cn + the breakpoint decode's reads)
            m.get_model_data()             # :286-297, the dir() walk
            # the properties analysis/pipeline.py:198-226 reads afterwards
            m.h; m.p_outlier_total; m.p_outlier_allele; m.total_likelihood_mask; m.allele_likelihood_mask
            m.get_likelihood_param_values(); m.breakpoint_prob(); m.p_breakpoint
    finally:
        bp.RemixtModel = real
    out = dict(rec.arrays)
    out['events'] = np.array(json.dumps(rec.events))
    out['meta'] = np.array(json.dumps({'case': name, 'N': N, 'M': M, 'max_cn': max_cn, 'normal_contamination': bool(normal_contamination),
                                       'fit_seed': fit_seed, 'num_em_iter': num_em_iter, 'num_update_iter': num_update_iter,
                                       'elbo': repr(float(m.prev_elbo))}))
    path = os.path.join(OUT, 'protocol_trace_%s.npz' % name)
    np.savez_compressed(path, **out)
    ops = {}
    for ev in rec.events:
        k = ev['op'] + (':' + ev['name'] if ev['op'] == 'call' else '')
        ops[k] = ops.get(k, 0) + 1
    print(name, 'events', len(rec.events), 'arrays', len(rec.arrays), 'bytes', os.path.getsize(path), 'elbo', m.prev_elbo)
    print('   ', sorted(ops.items()))


def main():
    build_ref.build()
    # two clones with normal contamination (four likelihood parameters)
    trace_case('m2', N=40, M=2, max_cn=4, chains=2, seed=1)
    # three clones WITHOUT normal contamination (ten likelihood parameters: hdel / LOH branches of the M-step)
    trace_case('m3_nonormal', N=48, M=3, max_cn=2, chains=2, seed=34, normal_contamination=False, zero_alleles=(4,))


if __name__ == '__main__':
    main()
