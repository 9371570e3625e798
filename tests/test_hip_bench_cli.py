"""bench.py's additional measurements run in child processes of their own (`--sub-run`, DESIGN 4.6: a process's first restart groups
find the hardware queues unused): the child's line at a small size, and the parent's view of it."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from remixt_amd import bpmodel
    return bpmodel


def test_sub_run_child_prints_one_parsable_line(hip):
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--sub-run', '4,2,8,2,1,1', '--segments', '3000'], capture_output=True, text=True, timeout=300)
    lines = [l for l in res.stdout.splitlines() if l.startswith('SUB_RUN ')]
    assert res.returncode == 0 and len(lines) == 1, (res.returncode, res.stdout[-300:], res.stderr[-300:])
    j = json.loads(lines[0][len('SUB_RUN '):])
    assert j['S'] == 165 and j['N1'] >= 3000 and j['dt'] > 0. and j['elbo_best'] < 0.
    assert j['info']['12'] == 1 and 1 <= j['info']['13'] <= j['info']['15'] <= 4      # k_fbm, restarts per workgroup
    assert 'k_fb' in j['prof'] and j['prof']['k_fb'][1] == 2 * 2 * 5                   # two groups x two timed EM iterations x five sweeps


def test_parent_reads_the_child_run(hip):
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(segments=3000, clones=3, update_iters=5, restarts=4, option=[], host_option=[], lib=None)
    rs, S, N1, dt, elbo, prof = bench._timed_run_isolated(args, 0, 4, 2, 8, 2, 1)
    assert isinstance(rs, bench._RunInfo) and rs.sets == [] and rs.paced is True      # (groups of two restarts are paced, RestartGroups paced='auto')
    assert S == 165 and dt > 0. and elbo.shape == (1,) and rs.batches[0].info(12) == 1 and prof['k_fb'][1] == 20
    bench._release(rs)                                                                # (nothing to release: a no-op)
