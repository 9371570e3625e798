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
