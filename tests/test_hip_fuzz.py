"""A fixed handful of tests/fuzz_small.py's randomised small, irregular problems (HIP path against the CPU oracle after every
coordinate update): ragged restart counts, chains of two segments, breakpoints on a third of the segments, every forward-backward
workgroup shape."""
import pytest

from tests import fuzz_small

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 5, 8, 13, 21, 34, 55])
def test_small_irregular_problem_matches_oracle(oracle_mod, seed):
    fuzz_small.run_case(fuzz_small.draw_case(seed), oracle_mod)


@pytest.mark.parametrize('seed', [6, 9, 13, 30, 53])
def test_small_irregular_problem_fits_like_the_oracle_driver(oracle_mod, seed):
    """Two whole EM iterations (sweeps, lock-step h M-step, shared-round searches, joint accept, ELBO) of the batched device driver against
    the per-restart driver over the oracle."""
    fuzz_small.run_fit_case(fuzz_small.draw_case(seed), oracle_mod)


@pytest.mark.parametrize('seed', [0, 3, 4, 7, 11, 59])
def test_large_grid_on_a_few_segments_matches_oracle(oracle_mod, seed):
    """The --huge mode (round 5): grids from 205 to 617 states and four clones at 207 / 457 on 10-24 segments -- k_fbq's and k_fbk's other shapes, ragged units,
    and the decode through lattice clusters and the parallel trace-back."""
    fuzz_small.run_case(fuzz_small.draw_case(seed, huge=True), oracle_mod)
