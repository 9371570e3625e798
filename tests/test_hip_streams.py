"""GPU: the process-wide stream pool.  A batch takes its two streams (sweeps / M-step; breakend branch of a sweep) from a
pool per device and role and gives them back when it is destroyed; pooled streams are never destroyed, so the hardware queue a role gets is
decided once per process and a later batch finds the placement the first ones found (profiles/r05_stream_pool.txt has the timings).  Here:
the pool is re-used, not grown, by batches built after others were closed; batches alive side by side get streams of their own; results do
not depend on whether a stream is fresh or re-used, pooled or private."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fit(options=None):
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    e = synthetic.make_experiment(600, num_clones=3, max_copy_number=4, num_chains=4, seed=51, num_breakpoints=10)
    ps = synthetic.make_init_params(e, 4, 4)
    rs = RestartGroups(e, ps, 4, groups=2, num_clones=3, quiet=True, seeds=[1, 2, 3, 4], options=options)
    elbo = rs.fit(num_em_iter=1, num_update_iter=2)
    return rs, elbo


def test_streams_are_pooled_reused_and_never_change_results():
    from remixt_amd import bpmodel
    rs1, e1 = _fit()
    b = rs1.batches[0]
    in_use = lambda bb: bb.info(16) - bb.info(17)
    base = in_use(b) - 4                                        # (streams of batches other tests of this process still hold)
    created = b.info(16)
    assert created >= 4 and in_use(b) == base + 4               # two groups x (main + breakend branch)
    rs1.close()
    rs2, e2 = _fit()                                            # after close(): the same streams again, none created
    b2 = rs2.batches[0]
    assert b2.info(16) == created and in_use(b2) == base + 4
    assert np.array_equal(e1, e2)
    rs3, e3 = _fit()                                            # next to a live pair of groups: four more streams in use (idle ones first, then new ones)
    b3 = rs3.batches[0]
    assert in_use(b3) == base + 8 and created <= b3.info(16) <= created + 4      # (idle streams of the right role first, then new ones)
    assert np.array_equal(e1, e3)
    rs2.close(); rs3.close()
    # private streams (created and destroyed per batch, round 4's behaviour): the pool is not touched, the fit is the same
    bpmodel.set_default_option('stream_pool', 0)
    try:
        rs4, e4 = _fit()
        assert rs4.batches[0].get_option('stream_pool') == 0 and rs4.batches[0].info(16) - rs4.batches[0].info(17) == base
        assert np.array_equal(e1, e4)
        rs4.close()
    finally:
        bpmodel.set_default_option('stream_pool', 1)
