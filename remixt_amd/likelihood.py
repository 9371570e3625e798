"""Read-count helpers of the reference's remixt/likelihood.py that sit next to the hot path
(SURVEY.md 8a row l1): used by read-depth initialisation and the result tables, not by the kernels."""
import numpy as np


class ProbabilityError(ValueError):
    def __init__(self, message, **variables):
        ValueError.__init__(self, message)
        self.variables = variables


# likelihood.py:67: rows = alleles (major, minor), columns = measurements (major, minor, total)
allele_measurement_matrix = np.array([[1, 0, 1], [0, 1, 1]])


def estimate_phi(x):
    """Proportion of genotypable reads per segment (likelihood.py:71-83): (major + minor) / (total + 1)."""
    x = np.asarray(x)
    return x[:, 0:2].sum(axis=1).astype(float) / (x[:, 2].astype(float) + 1.0)


def proportion_measureable_matrix(phi):
    """(N, 3) segment-to-measurement transform [phi, phi, 1] (likelihood.py:87-98)."""
    phi = np.asarray(phi)
    return np.vstack([phi, phi, np.ones(phi.shape)]).T


def expected_read_count(l, cn, h, phi):
    """Expected [major, minor, total] read counts l * sum_m h_m cn * [phi, phi, 1] + 1e-16
    (likelihood.py:101-134); raises ProbabilityError on a non-positive or nan mean."""
    l = np.asarray(l); cn = np.asarray(cn); h = np.asarray(h)
    p = proportion_measureable_matrix(phi)
    gamma = np.sum(cn * np.vstack([h, h]).T, axis=-2)         # (N, 2) allele depths
    x1 = np.dot(allele_measurement_matrix.T, gamma.T).T       # (N, 3)
    x3 = ((x1 * p).T * l.T).T
    x3 += 1e-16
    bad = np.argwhere(x3 <= 0)
    if len(bad):
        n = int(bad[0][0])
        raise ProbabilityError('mu <= 0', n=n, cn=cn[n], l=l[n], h=h, p=p[n], mu=x3[n])
    bad = np.argwhere(np.isnan(x3))
    if len(bad):
        n = int(bad[0][0])
        raise ProbabilityError('mu is nan', n=n, cn=cn[n], l=l[n], h=h, p=p[n], mu=x3[n])
    return x3


# ---------------------------------------------------------------------------------------------------
# The two count distributions of the hot path in vectorised form: the reference keeps them in
# likelihood.py next to the Cython scalars of bpmodel.pyx as an independent statement of the same log pmfs, and
# tests/test_distributions.py uses them the same way against the oracle's and the HIP kernels' per-cell values.
# Pinned by tests/golden/distributions.npz, recorded from the reference's classes.
# ---------------------------------------------------------------------------------------------------
def _lgamma(v):
    from scipy.special import gammaln
    return gammaln(v)


def _digamma(v):
    from scipy.special import digamma
    return digamma(v)


class NegBinDistribution(object):
    """Negative binomial over total read counts, mean `mu`, over-dispersion `r` (likelihood.py:569-662;
    scalar twin bpmodel.pyx:238-301)."""

    def __init__(self, **kwargs):
        self.r = kwargs.get('r', 500.)

    def _success_probability(self, mu):
        q = np.array(mu / (self.r + mu), dtype=float, copy=True, ndmin=1)
        q[(q < 0.) | (q > 1.)] = 0.5          # likelihood.py:601-602: out-of-range means are evaluated at 1/2 instead of failing
        return q

    def log_likelihood(self, x, mu):
        """log C(x + r - 1, x) + x log q + r log(1 - q) with q = mu / (r + mu), per segment."""
        x = np.asarray(x, dtype=float); mu = np.asarray(mu, dtype=float)
        q = self._success_probability(mu).reshape(np.shape(mu))
        combinatorial = _lgamma(x + self.r) - _lgamma(x + 1) - _lgamma(self.r)
        return combinatorial + x * np.log(q) + self.r * np.log(1 - q)

    def log_likelihood_partial_mu(self, x, mu):
        return x / mu - (self.r + x) / (self.r + mu)

    def log_likelihood_partial_r(self, x, mu):
        r = self.r
        rising = _digamma(r + x) - _digamma(r)
        return rising + np.log(r) + 1. - np.log(r + mu) - r / (r + mu) - x / (r + mu)


class BetaBinDistribution(object):
    """Beta binomial over minor-allele counts `k` of `n`, expected fraction `p`, precision `M`
    (likelihood.py:949-1084; scalar twin bpmodel.pyx:304-394)."""

    def __init__(self, **kwargs):
        self.M = kwargs.get('M', 500.)

    def _shape(self, p):
        return self.M * p, self.M * (1 - p)          # the two beta shape parameters

    def log_likelihood(self, k, n, p):
        """log C(n, k) + log B(k + a, n - k + b) - log B(a, b) with (a, b) = (M p, M (1 - p))."""
        a, b = self._shape(p)
        choose = _lgamma(n + 1) - _lgamma(k + 1) - _lgamma(n - k + 1)
        return (choose + _lgamma(k + a) + _lgamma(n - k + b) - _lgamma(n + self.M)
                - _lgamma(a) - _lgamma(b) + _lgamma(self.M))

    def log_likelihood_partial_p(self, k, n, p):
        a, b = self._shape(p)
        M = self.M
        return M * _digamma(k + a) + (-M) * _digamma(n - k + b) - M * _digamma(a) - (-M) * _digamma(b)

    def log_likelihood_partial_M(self, k, n, p):
        a, b = self._shape(p)
        return (p * _digamma(k + a) + (1 - p) * _digamma(n - k + b) - _digamma(n + self.M)
                - p * _digamma(a) - (1 - p) * _digamma(b) + _digamma(self.M))
