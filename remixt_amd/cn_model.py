"""Host side of the hot path: `BreakpointModel`, API-compatible with the
reference's `remixt.cn_model.BreakpointModel` (reference remixt/cn_model.py:29-628).

Segment remapping for breakends, likelihood masks, state-grid enumeration, the
EM loop and the scipy M-steps live here (Python, as in the reference); every
numeric kernel call goes to `remixt_amd.bpmodel.RemixtModel` (HIP).  The class
is written against the kernel *protocol* only, so tests can run the identical
host logic over the reference binary or the CPU oracle by passing
`kernel_module=`; the default and only product kernel is the HIP one -- there is
no fallback if it cannot be loaded.
"""
import collections
import contextlib
import datetime
import itertools

import numpy as np
import scipy.optimize


def _get_brkend_seg_orient(breakend):
    """Breakend (segment, side) -> (left segment of the adjacency, orientation)  (cn_model.py:14-22)."""
    n, side = breakend
    if side == 1:
        return n, +1
    elif side == 0:
        return n - 1, -1
    raise ValueError('side must be 0 or 1')


def _gettime():
    return datetime.datetime.now().time().isoformat()


def create_cn_states(num_clones, num_alleles, cn_max, cn_diff_max):
    """Allele-specific copy-number states of one segment (cn_model.py:228-253).

    The reference enumerates itertools.product in order and de-duplicates states
    that are equal under swapping the two alleles with a dict keyed by the
    unordered pair {state, swapped}: the resulting order is that of the FIRST
    occurrence of each key and the stored value is the LAST occurrence.  In
    product order a state precedes its swap iff its flattened tuple is
    lexicographically smaller, so: keep tuples with key <= swapped key, in
    product order, and emit max(key, swapped).
    """
    assert num_alleles == 2
    nv = (num_clones - 1) * num_alleles
    if nv == 0:
        return np.array([[[1, 1]]], dtype=np.int64)
    grids = np.indices((cn_max + 1,) * nv).reshape(nv, -1).T          # product order
    t = grids.reshape(-1, num_clones - 1, num_alleles)
    ok = (t.sum(axis=2) <= cn_max).all(axis=1)
    ok &= ((t.max(axis=1) - t.min(axis=1)) <= cn_diff_max).all(axis=1)
    t = t[ok]
    key = t.reshape(len(t), -1)
    swapped = t[:, :, ::-1].reshape(len(t), -1)
    diff = key != swapped
    first = np.argmax(diff, axis=1)
    rows = np.arange(len(t))
    key_lt = diff.any(axis=1) & (key[rows, first] < swapped[rows, first])
    key_eq = ~diff.any(axis=1)
    keep = key_lt | key_eq
    out = np.where(key_lt[:, None], swapped, key)[keep].reshape(-1, num_clones - 1, num_alleles)
    normal = np.ones((len(out), 1, num_alleles), dtype=out.dtype)
    return np.concatenate([normal, out], axis=1).astype(np.int64)


def create_brk_states(num_clones, cn_max, cn_diff_max):
    """Breakpoint copy-number states (cn_model.py:255-276)."""
    nv = num_clones - 1
    if nv == 0:
        return np.zeros((1, 1), dtype=np.int64)
    grids = np.indices((cn_max + 1,) * nv).reshape(nv, -1).T
    ok = (grids.max(axis=1) - grids.min(axis=1)) <= cn_diff_max
    grids = grids[ok]
    return np.concatenate([np.zeros((len(grids), 1), dtype=grids.dtype), grids], axis=1).astype(np.int64)


_NATIVE_SEARCH = [False, None]


def _native_weighted_search():
    """rmx_weighted_search of the HIP library if it is loadable (it is host code), else None: the numpy
    formulation gives the same indices."""
    if _NATIVE_SEARCH[0]:
        return _NATIVE_SEARCH[1]
    _NATIVE_SEARCH[0] = True
    try:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)

        def search(p, u):
            u = np.ascontiguousarray(u, dtype=np.float64)
            out = np.empty(len(u), dtype=np.int64)
            pos = C.c_int64(0)
            rc = lib.rmx_weighted_search(p.ctypes.data_as(dp), len(p), u.ctypes.data_as(dp), len(u), out.ctypes.data_as(ip), C.byref(pos))
            if rc != 0:
                raise RuntimeError('rmx_weighted_search failed')
            return out, int(pos.value)
        _NATIVE_SEARCH[1] = search
    except Exception:
        _NATIVE_SEARCH[1] = None
    return _NATIVE_SEARCH[1]


class WeightColumn(object):
    """Sample weights given as a column of an (N, 2) float64 indicator array plus their sum -- what get_param_sample_weight
    returns for negbin_r_* / betabin_M_* when asked not to materialise `column / norm`: _sample_without_replacement then runs
    each round of draws in one native call on the strided column (rmx_weighted_sample_round; same values, same draws)."""
    __slots__ = ('array', 'column', 'norm')

    def __init__(self, array, column, norm):
        self.array, self.column, self.norm = array, int(column), float(norm)

    def dense(self):
        return self.array[:, self.column] / self.norm


_NATIVE_ROUND = [False, None]


def _native_sample_round():
    """rmx_weighted_sample_round of the HIP library if it is loadable (host code, runs without the GIL), else None."""
    if _NATIVE_ROUND[0]:
        return _NATIVE_ROUND[1]
    _NATIVE_ROUND[0] = True
    try:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        fn = lib.rmx_weighted_sample_round
        dp, ip, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)

        def round_(wc, u, found, nfound, cap):
            """Append the new distinct indices the draws `u` select to found[:nfound]; returns (nfound, positive weights)."""
            a = wc.array
            if a.dtype != np.float64 or a.ndim != 2 or not a.flags.c_contiguous:
                raise TypeError('indicator array must be C-contiguous float64')
            nf = C.c_int32(nfound); pos = C.c_int64(0)
            base = C.cast(a.ctypes.data + 8 * wc.column, dp)
            rc = fn(base, a.shape[0], a.shape[1], wc.norm, u.ctypes.data_as(dp), len(u), found.ctypes.data_as(ip), C.byref(nf), cap, C.byref(pos))
            if rc != 0:
                raise RuntimeError('rmx_weighted_sample_round failed')
            return int(nf.value), int(pos.value)
        _NATIVE_ROUND[1] = round_
    except Exception:
        _NATIVE_ROUND[1] = None
    return _NATIVE_ROUND[1]


def _sample_without_replacement(rng, n, size, p=None):
    """`size` distinct indices from range(n), successively with probability proportional to p
    (uniform if None): the distribution of numpy's choice(n, size, replace=False, p=p).  numpy
    re-normalises and re-accumulates p after every batch of draws and permutes all n items in the
    uniform case; drawing WITH replacement from the one cumulative distribution and keeping first
    occurrences is the same process (a repeat is exactly a draw the renormalised distribution would
    not have produced) and needs one cumsum / no permutation.  p may be a WeightColumn (see there)."""
    if size > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    if isinstance(p, WeightColumn):
        round_ = _native_sample_round()
        if round_ is None:
            p = p.dense()
        else:
            # the same rounds as below -- draws, cumulative sum, binary search, first occurrences -- each round one native call
            found = np.empty(size + size // 2 + 8, dtype=np.int64)
            nfound = 0
            while nfound < size:
                k = size - nfound
                k += k // 2 + 8
                u = rng.rand(k)
                if len(found) < nfound + k:
                    found = np.concatenate([found[:nfound], np.empty(k, dtype=np.int64)])
                nfound, positive = round_(p, u, found, nfound, len(found))
                if positive < size:
                    raise ValueError("Fewer non-zero entries in p than size")
            return found[:size].copy()
    cdf = None
    native = _native_weighted_search() if p is not None else None
    if p is not None and native is None:
        if np.count_nonzero(p > 0) < size:
            raise ValueError("Fewer non-zero entries in p than size")
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
    if native is not None:
        p = np.ascontiguousarray(p, dtype=np.float64)
    found = np.empty(0, dtype=np.int64)
    while found.size < size:
        k = size - found.size
        k += k // 2 + 8
        if p is None:
            new = rng.randint(0, n, size=k).astype(np.int64)
        elif native is not None:
            # the same cumulative sum and binary search in C++ (rmx_weighted_search): no GIL, so the
            # restarts' draws really run side by side on the host threads
            new, positive = native(p, rng.rand(k))
            if positive < size:
                raise ValueError("Fewer non-zero entries in p than size")
        else:
            new = np.minimum(cdf.searchsorted(rng.rand(k), side='right'), n - 1).astype(np.int64)
        cand = np.concatenate([found, new])
        _, first = np.unique(cand, return_index=True)
        first.sort()
        found = cand.take(first)
    return found[:size]


class BreakpointModel(object):

    def __init__(self, x, l, adjacencies, breakpoints, **kwargs):
        """Create a copy number model (same arguments as cn_model.py:31-74).

        Extra keyword arguments: `kernel_module` (module providing RemixtModel;
        default remixt_amd.bpmodel), `device` (HIP device ordinal), `quiet`, `rng`.
        """
        x = np.asarray(x)
        l = np.asarray(l)
        # Observed data should be ordered as major, minor, total
        assert np.all(x[:, 1] <= x[:, 0])

        self.N = x.shape[0]
        # cn_model.py:59 -- raises on an empty dict, as the reference does
        self.breakpoint_ids, self.breakpoints = zip(*breakpoints.items())

        self.max_copy_number = kwargs.get('max_copy_number', 6)
        self.max_copy_number_diff = kwargs.get('max_copy_number_diff', 1)
        self.normal_contamination = kwargs.get('normal_contamination', True)
        self.is_female = kwargs.get('is_female', True)
        self.divergence_weight = kwargs.get('divergence_weight', 1e6)
        self.min_segment_length = kwargs.get('min_segment_length', 10000)
        self.min_proportion_genotyped = kwargs.get('min_proportion_genotyped', 0.01)
        self.max_depth = kwargs.get('max_depth')
        self.transition_log_prob = kwargs.get('transition_log_prob', 10.)
        self.transition_model = kwargs.get('transition_model', 0)
        self.disable_breakpoints = kwargs.get('disable_breakpoints', False)
        self.breakpoint_init = kwargs.get('breakpoint_init', None)
        self.normal_copies = kwargs.get('normal_copies', None)
        if self.normal_copies is None:
            self.normal_copies = np.ones((self.N, 2), dtype=np.int64)
        self.normal_copies = np.asarray(self.normal_copies)
        self.do_h_update = kwargs.get('do_h_update', True)
        self._kernel = kwargs.get('kernel_module', None)
        self._device = kwargs.get('device', 0)
        self.quiet = kwargs.get('quiet', False)
        # None: the global numpy RNG, as in the reference (cn_model.py:477); a RandomState gives a
        # restart its own reproducible stream when several restarts advance in lockstep
        self.rng = kwargs.get('rng', None)

        if self.max_depth is None:
            raise ValueError('must specify max depth')

        if not self.normal_contamination:
            self.normal_copies = self.normal_copies * 0

        # restarts of one experiment share the segment remap (same adjacencies and breakpoints):
        # the first model of a RestartSet computes it, the others copy the arrays
        remap_cache = kwargs.get('remap_cache', None)
        names = ('N1', 'seg_fwd_remap', 'seg_is_original', 'seg_rev_remap', 'num_breakpoints',
                 'is_telomere', 'breakpoint_idx', 'breakpoint_orient')
        if remap_cache is not None and 'remap' in remap_cache:
            for name, value in zip(names, remap_cache['remap']):
                setattr(self, name, value.copy() if isinstance(value, np.ndarray) else value)
        else:
            self._remap_segments(adjacencies)
            if remap_cache is not None:
                remap_cache['remap'] = tuple(getattr(self, name) for name in names)

        self.x1 = np.zeros((self.N1, x.shape[1]), dtype=float)
        self.l1 = np.zeros((self.N1,), dtype=float)
        self.x1[self.seg_fwd_remap, :] = x
        self.l1[self.seg_fwd_remap] = l

        # Likelihood masks (cn_model.py:169-184)
        long_enough = self.l1 >= self.min_segment_length
        genotyped = self.x1[:, :2].sum(axis=1).astype(float) / (self.x1[:, 2].astype(float) + 1e-16)
        depth = self.x1[:, 2].astype(float) / (self.l1.astype(float) + 1e-16)
        shallow = depth <= self.max_depth
        self._total_likelihood_mask = long_enough & shallow
        self._allele_likelihood_mask = long_enough & (genotyped >= self.min_proportion_genotyped) & shallow

        if self.disable_breakpoints:
            self.num_breakpoints = 0
            self.breakpoint_idx = -np.ones(self.breakpoint_idx.shape, dtype=int)
            self.breakpoint_orient = np.zeros(self.breakpoint_orient.shape, dtype=int)

        self.check_elbo = False
        self.prev_elbo = None
        self.prev_elbo_diff = None
        self.num_em_iter = 1
        self.num_update_iter = 1
        self.model = None

        self.likelihood_params = ['negbin_r_0', 'negbin_r_1', 'betabin_M_0', 'betabin_M_1']
        if not self.normal_contamination:
            self.likelihood_params.extend(['negbin_hdel_mu', 'negbin_hdel_r_0', 'negbin_hdel_r_1',
                                           'betabin_loh_p', 'betabin_loh_M_0', 'betabin_loh_M_1'])
        self.likelihood_param_bounds = {
            'negbin_r_0': (10., 2000.), 'negbin_r_1': (1., 2000.),
            'betabin_M_0': (10., 2000.), 'betabin_M_1': (1., 2000.),
            'negbin_hdel_mu': (1e-9, 1e-4), 'negbin_hdel_r_0': (10., 2000.), 'negbin_hdel_r_1': (1., 200.),
            'betabin_loh_p': (1e-5, 1e-2), 'betabin_loh_M_0': (10., 2000.), 'betabin_loh_M_1': (1., 200.),
        }

    # ------------------------------------------------------------------------------
    def _remap_segments(self, adjacencies):
        """Insert zero-length dummy segments so that at most one breakend lies between
        two consecutive model segments (cn_model.py:82-161).

        Between original segments n and n+1 the model gets one segment per breakend
        incident on that boundary (the first of them is the original segment n when
        n >= 0) plus, if (n, n+1) is not a reference adjacency, one extra telomere
        segment.  Breakends at one boundary are visited in the iteration order of a
        Python set of (bp_idx, be_idx, orient) tuples, as in the reference.
        """
        adjacencies = set(tuple(a) for a in adjacencies) if not isinstance(adjacencies, (set, frozenset)) else adjacencies
        at_boundary = collections.defaultdict(set)
        for bp_idx, breakpoint in enumerate(self.breakpoints):
            for be_idx, breakend in enumerate(breakpoint):
                n, orient = _get_brkend_seg_orient(breakend)
                at_boundary[n].add((bp_idx, be_idx, orient))

        # The walk of cn_model.py:96-147 over the boundaries n = -1 .. N-1, without a Python iteration per segment (43 ms of a fit's 74 ms of
        # construction at 50 000 segments): a boundary without breakends yields its original segment (n >= 0) and nothing else, so only the
        # boundaries that carry breakends (two per breakpoint) are visited one by one -- in the iteration order of their sets, as the reference does.
        N = self.N
        adjacent = np.fromiter(((n, n + 1) in adjacencies for n in range(-1, N)), dtype=bool, count=N + 1)      # index n + 1
        nbe = np.zeros(N + 1, dtype=int)
        for n in [n for n in at_boundary if n < -1 or n >= N]:
            del at_boundary[n]                # (a boundary the reference's walk never reaches: its breakends are missing below, and the count check says so)
        for n, ends in at_boundary.items():
            nbe[n + 1] = len(ends)
        count = np.where(nbe > 0, nbe + (~adjacent).astype(int), 1)
        if nbe[0] == 0:
            count[0] = 0                      # n = -1 without a breakend: no segment
        start = np.concatenate([[0], np.cumsum(count)])
        total = int(start[-1])
        seg_rev = np.repeat(np.arange(-1, N), count)
        is_orig = np.zeros(total, dtype=bool)
        is_tel = np.zeros(total, dtype=int)
        bidx = np.full(total, -1, dtype=int)
        borient = np.zeros(total, dtype=int)
        plain = (nbe == 0) & (count == 1)     # original segment alone: a telomere unless the reference adjacency follows
        is_orig[start[:-1][plain]] = True
        is_tel[start[:-1][plain]] = np.where(adjacent[plain], 0, 1)
        for n, ends in at_boundary.items():
            at = int(start[n + 1])
            for j, (bp_idx, be_idx, orient) in enumerate(ends):
                is_orig[at + j] = (j == 0 and n >= 0)
                bidx[at + j] = bp_idx; borient[at + j] = orient
            if not adjacent[n + 1]:
                is_tel[at + len(ends)] = 1
        fwd = start[1:N + 1].copy()           # the first segment a boundary n >= 0 yields is the original one in either case

        self.N1 = total
        self.seg_fwd_remap = fwd
        self.seg_is_original = np.array(is_orig, dtype=bool)
        # quirk kept: dummy segments created before segment 0 carry index -1 (cn_model.py:133)
        self.seg_rev_remap = np.array(seg_rev, dtype=int)
        self.num_breakpoints = len(self.breakpoints)
        self.is_telomere = np.array(is_tel, dtype=int)
        self.breakpoint_idx = np.array(bidx, dtype=int)
        self.breakpoint_orient = np.array(borient, dtype=int)

        assert not np.any((self.breakpoint_idx >= 0) & (self.is_telomere == 1))
        assert np.all(np.bincount(self.breakpoint_idx[self.breakpoint_idx >= 0]) == 2)

    # state grids: methods with the reference's signature (cn_model.py:228, :255)
    def create_cn_states(self, num_clones, num_alleles, cn_max, cn_diff_max):
        return create_cn_states(num_clones, num_alleles, cn_max, cn_diff_max)

    def create_brk_states(self, num_clones, cn_max, cn_diff_max):
        return create_brk_states(num_clones, cn_max, cn_diff_max)

    def _log(self, msg):
        if not self.quiet:
            print('[{}] {}'.format(_gettime(), msg))

    # ------------------------------------------------------------------------------
    def get_likelihood_param_values(self):
        return dict((name, getattr(self.model, name)) for name in self.likelihood_params)

    def get_model_data(self):
        """All non-callable attributes of the kernel model (cn_model.py:286-297)."""
        data = {}
        for a in dir(self.model):
            if a.startswith('_'):
                continue
            try:
                v = getattr(self.model, a)
            except (AttributeError, NotImplementedError):
                continue
            if callable(v):
                continue
            data[a] = np.asarray(v) if hasattr(v, 'shape') else v
        return data

    def _state_weights(self, flag):
        post = np.asarray(self.model.posterior_marginals)
        return (post * (np.asarray(flag) == 1)).sum(axis=-1)

    def _get_hdel_weights(self):
        return self._state_weights(self.model.is_hdel)

    def _get_loh_weights(self):
        return self._state_weights(self.model.is_loh)

    def get_param_sample_weight(self, name, as_column=False):
        """Segment weights for the stochastic M-step of one parameter (cn_model.py:323-352).  as_column: for the parameters
        weighted by one column of an outlier indicator array, that column and its sum (WeightColumn) instead of a normalised copy."""
        m = self.model
        # (RestartSet passes the outlier indicators it fetched once for the whole M-step: they do not change in it)
        cache = getattr(self, '_mstep_indicator_cache', None) or {}
        column = None
        if name in ('negbin_r_0', 'negbin_r_1'):
            q = cache.get('p_outlier_total')
            column = (np.asarray(m.p_outlier_total) if q is None else q)
            weights = column[:, int(name[-1])]
        elif name in ('betabin_M_0', 'betabin_M_1'):
            q = cache.get('p_outlier_allele')
            column = (np.asarray(m.p_outlier_allele) if q is None else q)
            weights = column[:, int(name[-1])]
        elif name == 'negbin_hdel_mu':
            weights = self._get_hdel_weights()
        elif name in ('negbin_hdel_r_0', 'negbin_hdel_r_1'):
            weights = self._get_hdel_weights() * np.asarray(m.p_outlier_total)[:, int(name[-1])]
        elif name == 'betabin_loh_p':
            weights = self._get_loh_weights()
        elif name in ('betabin_loh_M_0', 'betabin_loh_M_1'):
            weights = self._get_loh_weights() * np.asarray(m.p_outlier_allele)[:, int(name[-1])]
        else:
            raise KeyError(name)
        norm = weights.sum()
        if norm > 0.:
            if as_column and column is not None and column.dtype == np.float64 and column.ndim == 2 and column.flags.c_contiguous:
                return WeightColumn(column, int(name[-1]), norm)
            return weights / norm
        self._log('nothing for ' + name)
        return None

    # ------------------------------------------------------------------------------
    def _state_tables(self, M):
        """Class-compressed cn state tables: (classes (C,S,M,2), seg_class (N1,)).

        The reference builds cn_states[N1,S,M,2] = the grid with the normal row of each
        segment overwritten by normal_copies, then remapped by seg_rev_remap
        (cn_model.py:359-364; index -1 wraps to the last segment).
        """
        grid = self.create_cn_states(M, 2, self.max_copy_number, self.max_copy_number_diff)
        normal_rows = np.asarray(self.normal_copies)[self.seg_rev_remap]          # (N1, 2), -1 wraps like numpy
        # the distinct rows in lexicographic order and each segment's row index (np.unique(..., axis=0, return_inverse=True)): through one integer key per
        # row where the copies are non-negative (always, in practice) -- the row-wise form sorts 50 000 structured records in 13 ms
        nr_ = np.asarray(normal_rows)
        if nr_.ndim == 2 and nr_.shape[1] == 2 and nr_.size and np.issubdtype(nr_.dtype, np.integer) and nr_.min() >= 0:
            base = int(nr_[:, 1].max()) + 1
            keys, seg_class = np.unique(nr_[:, 0].astype(np.int64) * base + nr_[:, 1], return_inverse=True)
            uniq = np.stack([keys // base, keys % base], axis=1).astype(nr_.dtype)
        else:
            uniq, seg_class = np.unique(normal_rows, axis=0, return_inverse=True)
        classes = np.repeat(grid[None], len(uniq), axis=0)
        classes[:, :, 0, :] = uniq[:, None, :]
        return classes.astype(np.int64), np.asarray(seg_class).reshape(-1).astype(np.int32)

    def _kernel_module(self):
        if self._kernel is None:
            from . import bpmodel   # HIP backend; raises if the library is missing
            self._kernel = bpmodel
        return self._kernel

    def _build_model(self, h_init):
        M = h_init.shape[0]
        classes, seg_class = self._state_tables(M)
        brk_states = self.create_brk_states(M, self.max_copy_number, self.max_copy_number_diff)
        kern = self._kernel_module()
        common = (h_init, self.l1, self.x1[:, 2].copy(), self.x1[:, 0:2].copy(), self.is_telomere, self.breakpoint_idx,
                  self.breakpoint_orient, self.transition_log_prob, self.divergence_weight)
        if hasattr(kern.RemixtModel, 'from_classes'):
            return kern.RemixtModel.from_classes(M, self.N1, self.num_breakpoints, self.normal_contamination,
                                                 classes, seg_class, brk_states, *common, device=self._device)
        cn_states = classes[seg_class]
        return kern.RemixtModel(M, self.N1, self.num_breakpoints, self.normal_contamination, cn_states, brk_states, *common)

    def _attach_model(self, model):
        """Finish model set-up after construction (cn_model.py:386-404)."""
        self.model = model
        self.model.total_likelihood_mask = self._total_likelihood_mask.astype(int)
        self.model.allele_likelihood_mask = self._allele_likelihood_mask.astype(int)

        if self.breakpoint_init is not None:
            # (the reference reads `self.model.self.num_breakpoints` here, an AttributeError; intent kept)
            p_breakpoint = np.ones((self.model.num_breakpoints, self.model.num_brk_states))
            brk_states = np.array(self.model.brk_states)
            for k, bp in enumerate(self.breakpoints):
                cn = self.breakpoint_init[bp]
                for s in range(self.model.num_brk_states):
                    if np.all(cn == brk_states[s]):
                        p_breakpoint[k, s] = 1000.
            p_breakpoint /= np.sum(p_breakpoint, axis=-1)[:, np.newaxis]
            self.model.p_breakpoint = p_breakpoint

        self.model.transition_model = self.transition_model

    def fit(self, h_init):
        """Fit the model with a series of updates (cn_model.py:354-428)."""
        h_init = np.asarray(h_init, dtype=float)
        self._attach_model(self._build_model(h_init))

        if self.prev_elbo is None:
            self.prev_elbo = self.model.calculate_elbo()

        for i in range(self.num_em_iter):
            self.em_iteration(i)

    def em_iteration(self, i=0, skip_variational=False):
        """One EM iteration: variational sweeps, M-steps, ELBO (cn_model.py:409-428)."""
        if not skip_variational:
            for j in range(self.num_update_iter):
                self.variational_update()
        if self.do_h_update:
            self.em_update_h()
        self.em_update_params()
        self.record_elbo(self.model.calculate_elbo(), i)

    def record_elbo(self, elbo, i=0):
        self.prev_elbo_diff = elbo - self.prev_elbo
        self.prev_elbo = elbo
        self._log('completed iteration {}'.format(i))
        self._log('    elbo: {:.10f}'.format(self.prev_elbo))
        self._log('    elbo diff: {:.10f}'.format(self.prev_elbo_diff))
        self._log('    h = {}'.format(np.asarray(self.model.h)))
        for name, value in self.get_likelihood_param_values().items():
            self._log('    {} = {}'.format(name, value))

    @contextlib.contextmanager
    def elbo_check(self, name, threshold=-1e-6):
        self._log('optimizing {}'.format(name))
        if not self.check_elbo:
            yield
            return
        elbo_before = self.model.calculate_elbo()
        yield
        elbo_after = self.model.calculate_elbo()
        self._log('    elbo: {:.10f}'.format(elbo_after))
        self._log('    elbo diff: {:.10f}'.format(elbo_after - elbo_before))
        if elbo_after - elbo_before < threshold:
            raise Exception('elbo error for step {}!'.format(name))

    def variational_update(self):
        """Single update of all variational parameters (cn_model.py:444-460)."""
        for name, step in (('update_p_allele_swap', 'update_p_allele_swap'), ('p_cn', 'update_p_cn'),
                           ('p_breakpoint', 'update_p_breakpoint'), ('p_outlier_total', 'update_p_outlier_total'),
                           ('p_outlier_allele', 'update_p_outlier_allele')):
            with self.elbo_check(name):
                getattr(self.model, step)()

    def em_update_h(self):
        with self.elbo_check('h'):
            self.update_h()

    def em_update_params(self):
        for name in self.likelihood_params:
            with self.elbo_check(name):
                self.update_param(name)

    def _draw_sample_indices(self, weights=None):
        """The segments of a stochastic M-step sample (cn_model.py:475-480; global numpy RNG unless the model owns one)."""
        sample_size = int(min(200, self.model.num_segments / 10))
        if self.rng is not None:
            # private stream (RestartSet): the same distribution, drawn with one cumulative sum
            return _sample_without_replacement(self.rng, self.model.num_segments, sample_size, weights)
        if isinstance(weights, WeightColumn):
            weights = weights.dense()
        return np.random.choice(self.model.num_segments, size=sample_size, replace=False, p=weights)

    def _create_sample(self, weights=None):
        """Random subset of segments for the stochastic M-steps as the reference's 0 / 1 mask."""
        sample = np.zeros((self.model.num_segments,), dtype=int)
        sample[self._draw_sample_indices(weights)] = 1
        return sample

    def _all_segments(self):
        return np.ones((self.model.num_segments,), dtype=int)

    def update_h(self):
        """Update haploid depths by optimising the expected log likelihood (cn_model.py:482-531)."""
        model = self.model

        def nll(h):
            model.h = h
            return -model.calculate_expected_log_likelihood(sample)

        def nll_grad(h):
            model.h = h
            partial_h = np.zeros((model.num_clones,))
            model.calculate_expected_log_likelihood_partial_h(sample, partial_h)
            return -partial_h

        h_before = np.array(model.h, dtype=float)
        ell_before = model.calculate_expected_log_likelihood(self._all_segments())
        sample = self._create_sample()

        result = scipy.optimize.minimize(nll, np.array(model.h, dtype=float), method='L-BFGS-B', jac=nll_grad,
                                         bounds=[(1e-8, 10.)] * model.num_clones)
        self._validate_h_result(result, nll, nll_grad)

        model.h = result.x
        ell_after = model.calculate_expected_log_likelihood(self._all_segments())
        if ell_after < ell_before:
            self._log('h rejected, elbo before: {}, after: {}'.format(ell_before, ell_after))
            model.h = h_before
        else:
            model.h = result.x

    @staticmethod
    def _validate_h_result(result, nll, nll_grad):
        """cn_model.py:510-521: what the reference does with an unsuccessful L-BFGS-B run."""
        if result.success:
            return
        message = result.message.decode() if isinstance(result.message, bytes) else str(result.message)
        if message == 'ABNORMAL_TERMINATION_IN_LNSRCH':
            # cn_model.py:513-518: the reference cross-checks the gradient numerically and continues
            # (statsmodels' forward difference there; scipy's here)
            analytic = nll_grad(result.x)
            numerical = scipy.optimize.approx_fprime(result.x, nll, 1e-8)
            if not np.allclose(analytic, numerical, atol=2.):
                raise ValueError('gradiant error, analytic: {}, numerical: {}\n'.format(analytic, numerical))
        else:
            raise ValueError('optimization failed\n{}'.format(result))

    def update_param(self, name):
        """Update one likelihood parameter by brute-force + simplex search (cn_model.py:533-569)."""
        model = self.model
        bounds = self.likelihood_param_bounds[name]
        weights = self.get_param_sample_weight(name)

        def nll(value):
            assert value.shape == (1,)
            value = float(value[0])
            if value < bounds[0] or value > bounds[1]:
                return np.inf
            setattr(model, name, value)
            return -model.calculate_expected_log_likelihood(sample)

        value_before = getattr(model, name)
        ell_before = model.calculate_expected_log_likelihood(self._all_segments())
        sample = self._create_sample(weights)

        result_value = self._brute_1d(nll, name, bounds, sample)

        # quirk kept: the acceptance test looks at the LAST value scipy evaluated, not at the optimum
        ell_after = model.calculate_expected_log_likelihood(self._all_segments())
        if ell_after < ell_before:
            self._log('{} rejected, elbo before: {}, after: {}'.format(name, ell_before, ell_after))
            setattr(model, name, value_before)
        else:
            setattr(model, name, result_value)

    def _brute_1d(self, nll, name, bounds, sample, Ns=20):
        """scipy.optimize.brute(nll, ranges=[bounds], full_output=True)[0] for one parameter, as the
        reference calls it (cn_model.py:553-561): a 20-point inclusive grid, argmin, then a
        scipy.optimize.fmin polish started at the best grid point.  Same evaluation sequence as
        scipy's own brute; the only difference is that a kernel offering
        calculate_expected_log_likelihood_param_grid evaluates the grid in one device round trip."""
        grid = np.mgrid[bounds[0]:bounds[1]:complex(Ns)]
        grid_eval = getattr(self.model, 'calculate_expected_log_likelihood_param_grid', None)
        if grid_eval is not None:
            Jout = -np.asarray(grid_eval(name, grid, sample))
        else:
            Jout = np.array([nll(np.asarray(x).flatten()) for x in grid])
        xmin = grid[int(np.argmin(Jout.ravel()))]
        res = scipy.optimize.fmin(nll, xmin, args=(), full_output=1, disp=False)
        value = np.asarray(res[0])
        assert value.shape == (1,)
        return float(value[0])

    # ------------------------------------------------------------------------------
    def optimal_cn(self, cn=None):
        """Viterbi decode + per-breakpoint argmax (cn_model.py:571-598).  `cn` = an already decoded
        path of this model ([N][M][2], model segment order), e.g. from a batched infer_cn."""
        m = self.model
        if cn is None:
            cn = np.zeros((m.num_segments, m.num_clones, m.num_alleles), dtype=int)
            m.infer_cn(cn)

        brk_states = np.asarray(m.brk_states)
        bidx = np.asarray(m.breakpoint_idx); borient = np.asarray(m.breakpoint_orient)
        log_breakpoint_p = np.zeros((m.num_breakpoints, m.num_brk_states))
        # the reference's loops over (n, clone, brk state): every entry of log_breakpoint_p accumulates
        # its terms in the same (n, clone) order here -- the j-th breakend (in n order) of all
        # breakpoints at once, clone by clone
        pen = m.transition_penalty
        ns = np.nonzero(bidx[:-1] >= 0)[0]
        if len(ns) > 0:
            order = np.argsort(bidx[ns], kind='stable')
            ns = ns[order]; ks = bidx[ns]
            starts = np.nonzero(np.r_[True, ks[1:] != ks[:-1]])[0]
            rank = np.arange(len(ks)) - np.repeat(starts, np.diff(np.r_[starts, len(ks)]))
            for j in range(int(rank.max()) + 1):
                sel = rank == j
                n_j = ns[sel]; k_j = ks[sel]
                # (total copies at the breakend adjacencies only -- 2 % of the segments -- instead of a sum over the whole path)
                tot_a = cn[n_j, :, 0] + cn[n_j, :, 1]; tot_b = cn[n_j + 1, :, 0] + cn[n_j + 1, :, 1]
                for c in range(m.num_clones):
                    d = tot_a[:, c] - tot_b[:, c]
                    log_breakpoint_p[k_j] += (-pen * np.abs(d[:, None] - borient[n_j][:, None] * brk_states[None, :, c]))

        brk_cn = dict()
        best = log_breakpoint_p.argmax(axis=1) if m.num_breakpoints else []
        for k in range(m.num_breakpoints):
            brk_cn[self.breakpoint_ids[k]] = np.array(brk_states[best[k]])      # a copy: brk_states may be a view of the kernel model's buffer

        return cn[self.seg_fwd_remap], brk_cn

    def breakpoint_prob(self):
        return dict(zip(self.breakpoints, np.asarray(self.model.p_breakpoint)))

    @property
    def h(self):
        return np.asarray(self.model.h)

    @property
    def p_breakpoint(self):
        return np.asarray(self.model.p_breakpoint)

    @property
    def p_outlier_total(self):
        return np.asarray(self.model.p_outlier_total)[self.seg_fwd_remap]

    @property
    def p_outlier_allele(self):
        return np.asarray(self.model.p_outlier_allele)[self.seg_fwd_remap]

    @property
    def total_likelihood_mask(self):
        return np.asarray(self.model.total_likelihood_mask)[self.seg_fwd_remap]

    @property
    def allele_likelihood_mask(self):
        return np.asarray(self.model.allele_likelihood_mask)[self.seg_fwd_remap]


def decode_breakpoints_naive(cn, adjacencies, breakpoints):
    """Breakpoint copy number from total copy-number steps at the breakends (cn_model.py:631-687)."""
    tot = np.asarray(cn).sum(axis=-1)
    partner = dict()
    for seg_1, seg_2 in adjacencies:
        partner[(seg_1, 1)] = (seg_2, 0)
        partner[(seg_2, 0)] = (seg_1, 1)

    def flow(breakend):
        n, side = breakend
        across = tot[partner[breakend][0], :] if breakend in partner else 0
        return np.maximum(tot[n, :] - across, 0)

    brk_cn = dict()
    for breakpoint_id, breakpoint in breakpoints.items():
        (be_1, be_2) = breakpoint
        brk_cn[breakpoint_id] = np.minimum(flow(tuple(be_1)), flow(tuple(be_2)))
    return brk_cn
