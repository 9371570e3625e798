"""Replay of a recorded call trace of the reference's host class (tests/golden/protocol_trace_*.npz, written by
oracle/make_protocol_trace.py) against a kernel class with the RemixtModel protocol.  Used by the GPU test (HIP kernel object) and by
the CPU suite (the oracle's kernel object, which pins the replay logic and the oracle at once)."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
RTOL = 1e-6
# dense arrays whose entries span hundreds of nats / many decades: compared with an absolute floor in their own units
ATOL = {'posterior_marginals': 1e-12, 'joint_posterior_marginals': 1e-12, 'p_breakpoint': 1e-12, 'p_outlier_total': 1e-12, 'p_outlier_allele': 1e-12,
        'p_allele_swap': 1e-12, 'framelogprob': 1e-9, 'log_transmat': 1e-9, 'cached_log_transmat': 1e-9,
        # dE[ll]/dh near the optimum L-BFGS-B walks to: per-cell terms of magnitude 1e2 .. 1e3 cancel down to 1e-5, so an element is
        # held to 1e-6 of ITSELF only above this floor (observed: 1.5e-10 absolute between the two digamma / accumulation orders)
        'calculate_expected_log_likelihood_partial_h': 1e-8}


def _value(d, ref):
    if ref is None:
        return None
    if 'a' in ref:
        return d[ref['a']]
    t = ref['t']
    return {'bool': bool, 'int': int, 'float': float, 'str': str}[t](ref['v'])


def _same(name, got, want):
    if isinstance(want, (bool, int)) and not isinstance(want, float):
        assert int(got) == int(want), (name, got, want)
        return
    if isinstance(want, float):
        assert np.isclose(float(got), want, rtol=RTOL, atol=1e-300), (name, got, want)
        return
    got = np.asarray(got)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if want.dtype.kind in 'iub':
        assert np.array_equal(got, want), name
    else:
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL.get(name, 1e-300), err_msg=name)


def replay(kernel_cls, case, check_dir=True):
    d = np.load(os.path.join(GOLDEN, 'protocol_trace_%s.npz' % case))
    events = json.loads(str(d['events']))
    meta = json.loads(str(d['meta']))
    model = None
    counts = {}
    for k, ev in enumerate(events):
        op = ev['op']
        counts[op] = counts.get(op, 0) + 1
        if op == 'construct':
            # positional, bpmodel.pyx:461-476; the arrays are the reference's own (int64 / float64, C-contiguous)
            model = kernel_cls(*[_value(d, a) for a in ev['args']])
        elif op == 'set':
            v = _value(d, ev['value'])
            setattr(model, ev['name'], v.copy() if isinstance(v, np.ndarray) else v)          # e.g. `model.h = ndarray` (cn_model.py:486)
        elif op == 'get':
            _same(ev['name'], getattr(model, ev['name']), _value(d, ev['value']))
        elif op == 'getmethod':
            if ev['name'].startswith('__'):          # CPython / Cython object plumbing met by the dir() walk (__reduce__, __setstate__ ...)
                continue
            assert callable(getattr(model, ev['name'])), ev['name']
        elif op == 'dir':
            # get_model_data walks dir(model) (cn_model.py:286-297): every public name of the reference object is answered
            # (check_dir=False: the oracle's test double answers the names through getattr but lists only its own fields)
            have = set(dir(model)) | set(n for n in dir(type(model)) if not n.startswith('_'))
            names = [n for n in ev['names'] if n != 'sum_product_2paramtrans']
            missing = [n for n in names if (check_dir and n not in have) or not hasattr(model, n)]
            assert not missing, missing
        elif op == 'call':
            args = [_value(d, a) for a in ev['args']]
            args = [a.copy() if isinstance(a, np.ndarray) else a for a in args]
            fn = getattr(model, ev['name'])
            if ev.get('raises'):
                with pytest.raises({'ValueError': ValueError, 'AssertionError': AssertionError}.get(ev['raises'], Exception)):
                    fn(*args)
                continue
            ret = fn(*args)
            where = '%s (event %d)' % (ev['name'], k)
            if ev['ret'] is not None:
                _same(where, ret, _value(d, ev['ret']))
            for i, ref in ev['out'].items():                                                   # caller-provided outputs
                want = _value(d, ref)
                if ev['name'] == 'infer_cn':
                    assert np.array_equal(args[int(i)], want), where
                else:
                    _same(ev['name'] if ev['name'] in ATOL else where, args[int(i)], want)
        else:
            raise AssertionError(op)
    # the fixture really is a whole fit: sweeps, both M-steps, ELBO, decode, the attribute walk
    assert counts['construct'] == 1 and counts['dir'] == 1 and counts['call'] > 500 and counts['set'] > 500 and counts['get'] > 100
    assert np.isclose(model.calculate_elbo(), float(meta['elbo']), rtol=RTOL)
