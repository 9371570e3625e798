"""Restart fan-out on GPUs.

The reference runs one OS process per (h, divergence weight) initialisation
(`init_id` axis, reference remixt/workflow.py:329-340) and picks the best ELBO
afterwards (remixt/analysis/pipeline.py:253-264).  Here all restarts of one GPU
share a single `RemixtBatch` (data resident once, one launch per coordinate
update for all restarts) and restarts are sharded over GPUs, one process per
GPU, with ONE gather of the per-restart results at the end (RCCL over xGMI via
torch.distributed; no communication during EM).
"""
import os

import numpy as np

from .cn_model import BreakpointModel
from . import synthetic


class SampleList(object):
    """An M-step sample (cn_model.py:475-480) as the ascending list of its segments; converts to the reference's dense 0 / 1
    mask on demand (np.asarray).  The batched driver uploads lists (rmx_set_sample_lists) and never needs the 50 000-entry mask
    that BreakpointModel._create_sample builds; the object doubles as the identity token of RemixtBatch._use_sample."""
    __slots__ = ('indices', 'n', '_mask')

    def __init__(self, indices, n):
        self.indices, self.n, self._mask = indices, int(n), None

    def __array__(self, dtype=None, copy=None):
        if self._mask is None:
            self._mask = np.zeros((self.n,), dtype=int)
            self._mask[self.indices] = 1
        return self._mask if dtype is None else self._mask.astype(dtype)

    def __len__(self):
        return self.n


import threading as _threading
_creation_lock = _threading.Lock()
_CREATION_OPTIONS = {'cu_partition': 0}      # creation-time options a batch may be given through options= (name -> the library's default)


class RestartSet(object):
    """R restarts of one experiment advancing in lockstep on one device."""

    def __init__(self, experiment, init_params, max_copy_number, num_clones=3, device=0, quiet=True,
                 kernel_module=None, seeds=None, strict=False, mstep_threads=2, lockstep=True, native_search=True, sample_prep=True,
                 options=None, h_init=None, joint_accept=True, **model_kwargs):
        self.experiment = experiment
        self.native_search = native_search
        self.strict = strict
        self.mstep_threads = mstep_threads
        self.lockstep = lockstep
        self.sample_prep = sample_prep      # draw the M-step samples on a helper thread while the device works
        self.joint_accept = joint_accept    # accept tests of the parameters searched together from one pass over the cells
        self.error_messages = {}
        self.init_params = list(init_params)
        R = len(self.init_params)
        if R == 0:
            raise ValueError('no restarts given')
        max_depths = set(p['max_depth'] for p in self.init_params)
        if len(max_depths) != 1:
            # analysis/pipeline.py:82-86: one common max_depth so that objectives are comparable
            raise ValueError('all restarts must share max_depth')
        self.num_clones = num_clones
        self.models = []
        remap_cache = model_kwargs.pop('remap_cache', None)
        if remap_cache is None:
            remap_cache = dict()
        for i, p in enumerate(self.init_params):
            rng = np.random.RandomState(seeds[i]) if seeds is not None else None
            self.models.append(BreakpointModel(
                experiment.x, experiment.l, experiment.adjacencies, experiment.breakpoints,
                max_copy_number=max_copy_number, divergence_weight=p['divergence_weight'], max_depth=p['max_depth'],
                kernel_module=kernel_module, device=device, quiet=quiet, rng=rng, remap_cache=remap_cache, **model_kwargs))
        # initial haploid depths: analysis/pipeline.py:128-132 from each restart's init parameters unless given (R, M)
        if h_init is not None:
            self.h_init = np.asarray(h_init, dtype=float).reshape(R, num_clones)
        else:
            self.h_init = np.array([synthetic.h_init_from_params(p, num_clones) for p in self.init_params])
        m0 = self.models[0]
        kern = m0._kernel_module()
        self.batch = None
        if hasattr(kern, 'RemixtBatch'):
            classes, seg_class = m0._state_tables(num_clones)
            brk_states = m0.create_brk_states(num_clones, m0.max_copy_number, m0.max_copy_number_diff)
            # creation-time options of THIS batch (e.g. cu_partition): the library reads its process-wide defaults when a batch is created, so they are
            # set and restored under a lock (restart groups are built side by side)
            create_options = dict((k, options[k]) for k in list(options or {}) if k in _CREATION_OPTIONS)
            options = dict((k, v) for k, v in (options or {}).items() if k not in _CREATION_OPTIONS)
            with _creation_lock:
                for k, v in create_options.items():
                    kern.set_default_option(k, v)
                try:
                    self.batch = kern.RemixtBatch(
                        num_clones, m0.N1, m0.num_breakpoints, m0.normal_contamination, classes, seg_class, brk_states,
                        self.h_init, m0.l1, m0.x1[:, 2].copy(), m0.x1[:, 0:2].copy(), m0.is_telomere, m0.breakpoint_idx,
                        m0.breakpoint_orient, m0.transition_log_prob, [p['divergence_weight'] for p in self.init_params],
                        device=device)
                finally:
                    for k in create_options:
                        kern.set_default_option(k, _CREATION_OPTIONS[k])
            for name, value in (options or {}).items():      # tuning options of the batch (tests, A/B measurements)
                self.batch.set_option(name, value)
            for r, m in enumerate(self.models):
                m._attach_model(self.batch.model(r))
        else:
            # kernels without a batch object (CPU oracle in tests): one model per restart, same driver
            for r, m in enumerate(self.models):
                m._attach_model(m._build_model(self.h_init[r]))

    @property
    def num_restarts(self):
        return len(self.models)

    def calculate_elbo(self):
        if self.batch is not None:
            return self.batch.calculate_elbo()
        return np.array([m.model.calculate_elbo() for m in self.models])

    def variational_update(self, iters=1):
        if self.batch is not None:
            self.batch.variational_update(iters)
        else:
            for m in self.models:
                for _ in range(iters):
                    m.variational_update()

    def _finish_pending_elbo(self):
        """The ELBO a deferred em_iteration queued (calculate_elbo_begin): wait for it and record it (cn_model.py:420-428)."""
        pending, self._elbo_pending_iter = getattr(self, '_elbo_pending_iter', None), None
        if pending is None:
            return None
        elbo = self.batch.calculate_elbo_end()
        for m, e in zip(self.models, elbo):
            m.record_elbo(float(e), pending)
        return elbo

    def em_iteration(self, i=0, num_update_iter=5, defer_elbo=False):
        """cn_model.py:409-428 for every restart: batched variational sweeps, per-restart
        scipy M-steps, batched ELBO.  defer_elbo: the iteration's ELBO is only QUEUED on the device (nothing in the next iteration depends on
        it: it is recorded, cn_model.py:420-428) and fetched behind the next iteration's sweeps -- or by _finish_pending_elbo(); the call then
        returns None.  The half millisecond between the ELBO's last kernel and the next sweep's first (the wait, the host's bookkeeping) was
        idle time of the restart group's stream in every EM iteration."""
        import time
        from . import lockstep as _ls
        t_ = [time.perf_counter()]
        # Lock-step parameter search needs the batched device objective and a private RNG stream per
        # restart (so that the order in which restarts draw their samples does not matter).
        lockstep = (self.lockstep and self.batch is not None and hasattr(self.batch, 'expected_log_likelihood_batch') and
                    all(m.rng is not None for m in self.models) and not any(m.check_elbo for m in self.models))
        h_lockstep = lockstep and hasattr(self.batch, 'expected_log_likelihood_h_batch') and _ls.lbfgsb_available()
        # the h M-step's (unweighted) samples are the first draws of the iteration whatever the sweeps give:
        # a helper thread draws them while this thread waits on the sweeps
        self._h_prefetch = None
        if h_lockstep and any(m.do_h_update for m in self.models) and self.sample_prep:
            if getattr(self, '_prep_pool', None) is None:
                from concurrent.futures import ThreadPoolExecutor
                self._prep_pool = ThreadPoolExecutor(max_workers=1)
            self._h_prefetch = ([m.rng.get_state() for m in self.models], self._prep_pool.submit(self._samples_and_lists))
        self.variational_update(num_update_iter)
        self._finish_pending_elbo()          # (the previous iteration's, if it was deferred: long finished behind these sweeps)
        t_.append(time.perf_counter())

        def mstep(r, with_h=True):
            m = self.models[r]
            if with_h and m.do_h_update:
                h_before = np.array(m.model.h, dtype=float)
                try:
                    m.em_update_h()
                except ValueError as err:
                    # the reference lets a failed L-BFGS-B run kill the whole restart job
                    # (cn_model.py:510-521); here the restart keeps its previous h and the
                    # message is reported in stats['error_message']
                    if self.strict:
                        raise
                    m.model.h = h_before
                    self.error_messages[r] = str(err).splitlines()[0] + ' (h kept)'
            if not lockstep:
                m.em_update_params()

        # The M-steps are host-latency-bound (hundreds of tiny objective evaluations per restart,
        # each a device round trip).  Preferred: every optimiser of every restart advances in lock
        # step and each round of evaluations is one batched device call.  Otherwise restarts run on
        # host threads so one restart's round trip overlaps the others' Python.  Both need a private
        # RNG stream per restart (seeds=...); with the reference's global numpy RNG the restarts run
        # one after the other.
        h_done = False
        if h_lockstep:
            h_done = self._update_h_lockstep()
        threaded = (self.batch is not None and self.mstep_threads > 1 and
                    all(m.rng is not None for m in self.models))
        if h_done and lockstep:
            pass
        elif threaded:
            list(self._threads().map(lambda r: mstep(r, not h_done), range(len(self.models))))
        else:
            for r in range(len(self.models)):
                mstep(r, not h_done)
        t_.append(time.perf_counter())
        if lockstep:
            self._update_params_lockstep()
        t_.append(time.perf_counter())
        if defer_elbo and self.batch is not None and hasattr(self.batch, 'calculate_elbo_begin'):
            self.batch.calculate_elbo_begin()
            self._elbo_pending_iter = i
            t_.append(time.perf_counter())
            self.phase_times = t_
            return None
        elbo = self.calculate_elbo()
        t_.append(time.perf_counter())
        self.phase_times = t_      # [start, after sweeps, after h M-step, after parameter M-steps, after ELBO]
        for m, e in zip(self.models, elbo):
            m.record_elbo(float(e), i)
        return elbo

    def _mark(self, label):
        """Host-side time stamps of the M-step's stages (tools/mstep_marks.py): off unless `self.marks` is a list."""
        marks = getattr(self, 'marks', None)
        if marks is not None:
            import time
            marks.append((label, time.perf_counter()))

    def _threads(self):
        if getattr(self, '_pool', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=max(1, self.mstep_threads))
        return self._pool

    def _samples(self, weights=None):
        """One M-step sample per restart (BreakpointModel._create_sample).  Every restart draws from
        its own RNG stream, so the draws run on host threads (numpy's cumsum / searchsorted /
        permutation release the GIL)."""
        def draw(r):
            return self.models[r]._create_sample(None if weights is None else weights[r])
        R = len(self.models)
        if self.mstep_threads > 1 and R > 1:
            return list(self._threads().map(draw, range(R)))
        return [draw(r) for r in range(R)]

    def _samples_and_lists(self):
        """One unweighted M-step sample per restart (the draws of _samples()), as SampleLists and as their ascending index lists
        (the form rmx_set_sample_lists uploads in one transfer)."""
        lists = [np.sort(m._draw_sample_indices(None)).astype(np.int32) for m in self.models]
        return [SampleList(lst, m.model.num_segments) for lst, m in zip(lists, self.models)], lists

    _MULTI_PARAMS = ('negbin_r_0', 'negbin_r_1', 'betabin_M_0', 'betabin_M_1')

    def _multi_param_names(self):
        """The leading standard likelihood parameters, whose searches can share their rounds."""
        b = self.batch
        first = []
        for name in self.models[0].likelihood_params:
            if name not in self._MULTI_PARAMS or len(first) == 4:
                break
            first.append(name)
        sequential = b is not None and hasattr(b, 'get_option') and b.get_option('search_mode') not in (0, 5, 6, 7)
        if not (first and self.native_search and b is not None and hasattr(b, 'param_search_multi')) or sequential:
            return []
        return first

    def _draw_param_samples(self, names):
        """Weighted samples of the listed parameters, parameter by parameter in the reference's order, and
        the outlier indicators they were weighted with: (samples {name: [per restart]}, indicators)."""
        fetched = self.batch.fetch_indicators() if (self.batch is not None and hasattr(self.batch, 'fetch_indicators')) else None

        def one(r):
            # a restart's indicators, then its draws for all listed parameters in order (its own RNG stream)
            m = self.models[r]
            if fetched is not None:
                c = {'p_outlier_total': fetched[0][r], 'p_outlier_allele': fetched[1][r]}      # views, valid for this M-step
            else:
                c = {'p_outlier_total': np.asarray(m.model.p_outlier_total), 'p_outlier_allele': np.asarray(m.model.p_outlier_allele)}
            m._mstep_indicator_cache = c
            # (the draws of m._create_sample(m.get_param_sample_weight(name)), without the normalised weight copies and the masks)
            lists = [np.sort(m._draw_sample_indices(m.get_param_sample_weight(name, as_column=True))).astype(np.int32) for name in names]
            return c, [SampleList(lst, m.model.num_segments) for lst in lists], lists
        R = len(self.models)
        if self.mstep_threads > 1 and R > 1:
            per_restart = list(self._threads().map(one, range(R)))
        else:
            per_restart = [one(r) for r in range(R)]
        ind = [c for c, _, _ in per_restart]
        samples = dict((name, [smp[j] for _, smp, _ in per_restart]) for j, name in enumerate(names))
        self._param_sample_lists = dict((name, [lst[j] for _, _, lst in per_restart]) for j, name in enumerate(names))
        return samples, ind

    def _start_param_sample_prep(self):
        self._param_prep = None
        names = self._multi_param_names()
        if not names or not self.sample_prep:
            return
        if getattr(self, '_prep_pool', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._prep_pool = ThreadPoolExecutor(max_workers=1)
        self._param_prep = (names, self._prep_pool.submit(self._draw_param_samples, names))

    def _drop_param_sample_prep(self):
        prep, self._param_prep = getattr(self, '_param_prep', None), None
        if prep is not None:
            try:
                prep[1].result()
            except Exception:
                pass

    def _update_h_lockstep(self):
        """BreakpointModel.update_h (cn_model.py:482-531) for all restarts at once: per restart the
        evaluation sequence of scipy's L-BFGS-B driver (remixt_amd/lockstep.py lbfgsb_gen), every
        round of (objective, gradient) evaluations one batched device call.  Returns False (state
        restored) when the batched path hit a device-side error, so that the caller can run the
        per-restart path, whose error handling is per restart."""
        from . import lockstep
        b = self.batch
        R = len(self.models)
        # a restart whose h M-step failed once is out of the selection for good (error_messages is never cleared):
        # it keeps its h from then on instead of failing -- and forcing a second lock-step run -- in every later iteration
        gone = getattr(self, '_h_failed', None)
        if gone is None:
            gone = self._h_failed = set()
        active = [r for r, m in enumerate(self.models) if m.do_h_update and r not in gone]
        if not active:
            return True
        h_before = [np.array(m.model.h, dtype=float) for m in self.models]
        prefetch, self._h_prefetch = getattr(self, '_h_prefetch', None), None
        rng_state = prefetch[0] if prefetch is not None else [m.rng.get_state() for m in self.models]
        dead = {}      # restart -> message: its objective raised one of the reference's ValueErrors during the search
        try:
            self._mark('h:start')
            ell_before = b.expected_log_likelihood_full(0, R)
            self._mark('h:ell_before')
            samples, sample_lists = prefetch[1].result() if prefetch is not None else self._samples_and_lists()
            self._mark('h:samples')
            # the parameter M-steps' samples come next in every restart's RNG stream and depend on the outlier
            # indicators only: they are drawn on a helper thread while this thread waits on the h rounds
            self._start_param_sample_prep()
            bounds = [(1e-8, 10.)] * b.num_clones
            while True:
                live = [r for r in active if r not in dead]
                if hasattr(b, 'set_sample_lists'):
                    b.set_sample_lists([(r, -1, samples[r], sample_lists[r]) for r in live])      # one transfer
                else:
                    for r in live:
                        b._use_sample(r, samples[r])
                self._mark('h:use_sample')

                if hasattr(b, 'h_batch_evaluator'):
                    evaluate = b.h_batch_evaluator(live)
                else:
                    def evaluate(ids, xs):
                        f, g = b.expected_log_likelihood_h_batch([live[i] for i in ids], np.asarray(xs, dtype=float))
                        return -np.asarray(f), -np.asarray(g)
                try:
                    # every restart's L-BFGS-B run (scipy's own routine, its evaluation sequence) in shared rounds
                    results = lockstep.lbfgsb_lockstep([h_before[r] for r in live], bounds, evaluate) if live else []
                    break
                except ValueError as err:
                    # The reference raises inside that restart's own process (e.g. total_depth <= 0 at a trial h) and only
                    # that restart dies; the batched call names the restarts it flagged.  They keep their h, the others
                    # start their (deterministic) optimiser runs again without them.
                    flagged = [r for r in getattr(err, 'restarts', []) if r in live]
                    if not flagged or self.strict:
                        raise
                    for r in flagged:
                        dead[r] = str(err).splitlines()[0] + ' (h kept)'
                    for r in live:
                        self.models[r].model.h = h_before[r]
        except ValueError:
            self._drop_param_sample_prep()
            for r, m in enumerate(self.models):
                m.model.h = h_before[r]
                m.rng.set_state(rng_state[r])
            return False
        self._mark('h:rounds')
        self.error_messages.update(dead)
        gone.update(dead)
        active = [r for r in active if r not in dead]
        failed = set()
        trial = hasattr(b, 'expected_log_likelihood_full_trial')
        for r, res in zip(active, results):
            m = self.models[r]
            try:
                if not res.success:
                    def nll(h, m=m, r=r):
                        m.model.h = h
                        return -m.model.calculate_expected_log_likelihood(samples[r])

                    def nll_grad(h, m=m, r=r):
                        m.model.h = h
                        out = np.zeros((b.num_clones,))
                        m.model.calculate_expected_log_likelihood_partial_h(samples[r], out)
                        return -out
                    m._validate_h_result(res, nll, nll_grad)
                m.model.h = res.x
            except ValueError as err:
                if self.strict:
                    raise
                failed.add(r)
                gone.add(r)
                self.error_messages[r] = str(err).splitlines()[0] + ' (h kept)'
        # accept test on trial values; a restart that keeps its h is rolled back without a second pass
        # over the cells (its expectations and cell cache still belong to h_before)
        self._mark('h:validate')
        ell_after = b.expected_log_likelihood_full_trial(0, R) if trial else None
        self._mark('h:trial')
        for r in failed:
            if trial:
                b.rollback_h(r, h_before[r])
            else:
                self.models[r].model.h = h_before[r]
        if not trial:
            ell_after = b.expected_log_likelihood_full(0, R)
        accepted = [False] * R
        for r in active:
            if r in failed:
                continue
            m = self.models[r]
            if ell_after[r] < ell_before[r]:
                m._log('h rejected, elbo before: {}, after: {}'.format(ell_before[r], ell_after[r]))
                if trial:
                    b.rollback_h(r, h_before[r])
                else:
                    m.model.h = h_before[r]
            else:
                accepted[r] = True
        # (the scratch expectations of the trial pass describe the state of every restart only if every restart kept its trial h)
        self._h_trial_kept = accepted if (trial and len(active) == R) else None
        self._mark('h:accept')
        return True

    def _update_params_lockstep(self):
        """BreakpointModel.em_update_params (cn_model.py:468-473, 533-569) for all restarts at once:
        per restart exactly the evaluation sequence of update_param -- full-data E[ll], weighted
        sample, 20-point grid, Nelder-Mead polish, full-data E[ll], accept / reject -- but every round
        of evaluations is one batched device call (remixt_amd/lockstep.py)."""
        from . import lockstep
        b = self.batch
        R = len(self.models)
        ids_all = list(range(R))
        # the outlier indicators feed the sample weights of several parameters and do not change during the
        # M-step: one device-to-host copy per restart and array instead of one per parameter (already made,
        # with the samples of the standard parameters, if the h M-step started the preparation)
        prep, self._param_prep = getattr(self, '_param_prep', None), None
        self._prepared = None
        self._mark('p:start')
        if prep is not None:
            samples, ind = prep[1].result()
            self._mark('p:prep_wait')
            self._prepared = (prep[0], samples)
            for m, c in zip(self.models, ind):
                m._mstep_indicator_cache = c
        else:
            for m in self.models:
                m._mstep_indicator_cache = {'p_outlier_total': np.asarray(m.model.p_outlier_total), 'p_outlier_allele': np.asarray(m.model.p_outlier_allele)}
        try:
            self._params_lockstep_body(b, R, ids_all)
        finally:
            self._prepared = None
            for m in self.models:
                m._mstep_indicator_cache = None

    def _search_standard_params_together(self, b, R, ids_all, names):
        """The searches of the leading standard parameters in shared evaluation rounds
        (rmx_param_search_multi): {name: (xopt, lastval)} or {} when the batch cannot.  The samples are
        drawn here, parameter by parameter in the reference's order, so every restart's RNG stream is
        consumed exactly as by the sequential loop (weights depend on the outlier indicators only)."""
        first = self._multi_param_names()
        if not first:
            return {}, {}
        bounds = [self.models[0].likelihood_param_bounds[name] for name in first]
        prepared = getattr(self, '_prepared', None)
        lists = None
        if prepared is not None and prepared[0] == first:
            samples = prepared[1]                  # drawn during the h M-step
            lists = getattr(self, '_param_sample_lists', None)
        else:
            samples = {}
            for name in first:
                samples[name] = self._samples([m.get_param_sample_weight(name) for m in self.models])
        self._mark('p:samples')
        if hasattr(b, 'set_sample_lists'):
            if lists is None or any(name not in lists for name in first):
                lists = dict((name, [np.flatnonzero(s).astype(np.int32) for s in samples[name]]) for name in first)
            b.set_sample_lists([(r, j, smp, lists[name][r]) for j, name in enumerate(first) for r, smp in enumerate(samples[name])])
        else:
            for j, name in enumerate(first):
                for r, smp in enumerate(samples[name]):
                    b.set_sample_slot(r, j, smp)
        self._mark('p:set_slots')
        grids = np.array([np.mgrid[lo:hi:complex(20)] for lo, hi in bounds])
        try:
            xopt, last = b.param_search_multi(ids_all, first, [lo for lo, hi in bounds], [hi for lo, hi in bounds], grids)
        except NotImplementedError:
            return {}, samples           # the samples are drawn: the sequential searches below use them
        self._mark('p:search')
        return dict((name, (xopt[j], last[j])) for j, name in enumerate(first)), samples

    def _accept_standard_params_together(self, b, R, ids_all, lead, together):
        """The accept tests of the parameters searched together (cn_model.py:563-569, one update_param after the other) from ONE
        pass over the cells: the parameters move disjoint components of the full-data E[ll], so E[ll] with parameter j at
        the last point its optimiser evaluated and the earlier parameters decided is a sum of component values at the
        committed and at the tried parameters (rmx_expected_ll_components; differs from the four full sums by rounding)."""
        comp = b.PARAM_COMPONENT
        value_before = dict((name, [b.get_param(r, name) for r in ids_all]) for name in lead)
        # E[ll] components before the parameter tests = at (h as decided, committed parameters).  Where the h M-step just ACCEPTED a new
        # h, its accept test's trial pass left exactly these expectations in scratch: summing them costs no pass over the cells, and the
        # restart's own expectations are refreshed once, after the parameter M-steps (by the ELBO), instead of here and there.
        kept = getattr(self, '_h_trial_kept', None)
        self._h_trial_kept = None
        cur = None
        if kept is not None and len(kept) == R:
            try:
                # (restarts that kept their trial h: from the trial pass's scratch; rolled back ones: from their own, still current)
                cur = b.expected_log_likelihood_components(0, R, trial=2 if all(kept) else 3)
            except NotImplementedError:
                cur = None
        if cur is None:
            cur = b.expected_log_likelihood_components(0, R)
        self._mark('p:ell_before')
        for name in lead:
            last = together[name][1]
            for r in ids_all:
                b.set_param(r, name, float(last[r]))          # where the sequential search leaves it
        tried = b.expected_log_likelihood_components(0, R, trial=True)
        self._mark('p:trial')

        def total(v):
            return (v[0] + v[1]) + (v[2] + v[3])
        for name in lead:
            c = comp[name]
            xopt = together[name][0]
            for r, m in enumerate(self.models):
                ell_before = total(cur[r])
                after = np.array(cur[r]); after[c] = tried[r, c]
                ell_after = total(after)
                if ell_after < ell_before:
                    m._log('{} rejected, elbo before: {}, after: {}'.format(name, ell_before, ell_after))
                    b.rollback_param(r, name, value_before[name][r])
                else:
                    b.set_param(r, name, float(xopt[r]))
                    cur[r, c] = tried[r, c]
        self._mark('p:accept')

    def _params_lockstep_body(self, b, R, ids_all):
        names = list(self.models[0].likelihood_params)
        together, drawn = self._search_standard_params_together(b, R, ids_all, names)
        lead = [name for name in names if name in together]
        if lead and self.joint_accept and hasattr(b, 'expected_log_likelihood_components') and names[:len(lead)] == lead:
            self._accept_standard_params_together(b, R, ids_all, lead, together)
            names = names[len(lead):]
        for name in names:
            lo, hi = self.models[0].likelihood_param_bounds[name]
            value_before = [b.get_param(r, name) for r in ids_all]
            ell_before = b.expected_log_likelihood_full(0, R)
            self._mark('p:ell_before')
            if name in together:
                # searched already, model untouched: put the parameter where the sequential search leaves it
                # (the last point the optimiser evaluated, cn_model.py:563-569)
                xopt, last = together[name]
                for r in ids_all:
                    b.set_param(r, name, float(last[r]))
            else:
                if name in drawn:
                    smps = drawn[name]
                else:
                    smps = self._samples([m.get_param_sample_weight(name) for m in self.models])
                for r, smp in enumerate(smps):
                    b._use_sample(r, smp)
                grid = np.mgrid[lo:hi:complex(20)]
                if self.native_search and hasattr(b, 'param_search'):
                    xopt = b.param_search(ids_all, name, lo, hi, grid)       # the same search, host loop in C++
                else:
                    xopt = self._param_search_python(name, lo, hi, grid)
            # the accept test on trial values: a rejected value is rolled back without a second pass over
            # the cells (the restart's expectations and cell cache still belong to value_before)
            trial = hasattr(b, 'expected_log_likelihood_full_trial')
            self._mark('p:set_trial')
            ell_after = b.expected_log_likelihood_full_trial(0, R) if trial else b.expected_log_likelihood_full(0, R)
            self._mark('p:trial')
            for r, m in enumerate(self.models):
                if ell_after[r] < ell_before[r]:
                    m._log('{} rejected, elbo before: {}, after: {}'.format(name, ell_before[r], ell_after[r]))
                    if trial:
                        b.rollback_param(r, name, value_before[r])
                    else:
                        b.set_param(r, name, value_before[r])
                else:
                    b.set_param(r, name, float(xopt[r]))
            self._mark('p:accept')

    def _param_search_python(self, name, lo, hi, grid):
        """The search of rmx_param_search driven from Python generators (remixt_amd/lockstep.py)."""
        from . import lockstep
        b = self.batch
        R = len(self.models)
        ids_all = list(range(R))
        J = np.empty((R, len(grid)))
        for gi, gv in enumerate(grid):
            J[:, gi] = -b.expected_log_likelihood_batch(ids_all, name, np.full(R, gv))
        xmin = grid[np.argmin(J, axis=1)]

        def evaluate(ids, xs):
            vals = [float(x[0]) for x in xs]
            out = [np.inf] * len(ids)          # outside the bounds: inf, model untouched (cn_model.py:542-543)
            sel = [k for k, v in enumerate(vals) if not (v < lo or v > hi)]
            if sel:
                res = b.expected_log_likelihood_batch([ids[k] for k in sel], name, [vals[k] for k in sel])
                for k, e in zip(sel, res):
                    out[k] = -float(e)
            return out
        results = lockstep.run_lockstep([lockstep.fmin_1d(xmin[r]) for r in ids_all], evaluate)
        return np.array([float(res[0][0]) for res in results])

    def close(self):
        """Destroy the device batch now (its memory, streams and events) instead of whenever the last reference goes."""
        try:
            self._finish_pending_elbo()
        except Exception:
            pass
        b, self.batch = self.batch, None
        for m in self.models:
            m.model = None
        if b is not None and hasattr(b, 'close'):
            b.close()

    def fit(self, num_em_iter=5, num_update_iter=5):
        elbo0 = self.calculate_elbo()
        for m, e in zip(self.models, elbo0):
            if m.prev_elbo is None:
                m.prev_elbo = float(e)
        for i in range(num_em_iter):
            self.em_iteration(i, num_update_iter, defer_elbo=i + 1 < num_em_iter)      # (the last one waits for its ELBO)
        self._finish_pending_elbo()
        return np.array([m.prev_elbo for m in self.models])

    def results(self):
        """Per-restart result dicts with the keys of analysis/pipeline.py:198-226."""
        self._finish_pending_elbo()
        out = []
        cn_all = None
        ind = None
        if self.batch is not None and len(self.models) > 0:
            cn_all, _ = self.batch.infer_cn_batch(0, len(self.models))     # all lattices side by side
            if hasattr(self.batch, 'fetch_indicators'):
                ind = self.batch.fetch_indicators()                        # the outlier indicators of all restarts in one transfer (views: copied below)
        for r, (m, p) in enumerate(zip(self.models, self.init_params)):
            res = collect_fit_results(m, self.experiment, p, cn=None if cn_all is None else cn_all[r],
                                      indicators=None if ind is None else (ind[0][r], ind[1][r]))
            res['stats']['error_message'] = self.error_messages.get(r, '')
            out.append(res)
        return out


class RestartGroups(object):
    """The restarts of one GPU split into `groups` RestartSets, each with its own device batch
    (own HIP stream) and its own host thread.

    One EM iteration alternates a device-bound phase (variational sweeps) with a host-latency-bound
    phase (scipy / Nelder-Mead M-steps whose objective evaluations are ~100 us device round trips).
    With two or more groups the phases of different groups overlap: while one group's host thread
    drives its M-step, the other group's sweeps keep the CUs busy, and kernels of different groups
    run concurrently on the device (a 184-workgroup forward-backward launch leaves CUs idle).
    Restarts are independent (reference remixt/workflow.py:329-340) and every restart owns its RNG
    stream, so a restart's fit does not depend on which group it runs in -- to the bit PER search mode and
    forward-backward workgroup shape.  The shape follows the grouping unless the caller pins it (`options=`):
    the library picks `fb_nv` by launch size.  (The search mode did too until round 4; the device-driven
    rounds, `search_mode` 5, are every batch's default now.)  Across shapes and search modes results agree
    to rounding (posteriors 1e-10; after an EM iteration's optimisers ELBO 1e-7, h 1e-5, parameters 1e-3),
    not to the bit."""

    def __init__(self, experiment, init_params, max_copy_number, groups=2, seeds=None, paced='auto', **kwargs):
        init_params = list(init_params)
        R = len(init_params)
        groups = max(1, min(int(groups), R))
        if seeds is None and groups > 1:
            raise ValueError('grouped restarts need one RNG seed per restart')
        bounds = [(R * g) // groups for g in range(groups + 1)]
        self.slices = [slice(bounds[g], bounds[g + 1]) for g in range(groups)]
        remap_cache = dict()

        # cu_partition=True: every group's streams get their own range of the device's CUs (library option cu_partition; 2, 4 or 8 groups)
        cu_partition = bool(kwargs.pop('cu_partition', False)) and groups in (2, 4, 8)

        def build(sl):
            kw = dict(kwargs)
            if cu_partition:
                kw['options'] = dict(kw.get('options') or {}, cu_partition=groups * 16 + self.slices.index(sl))
            return RestartSet(experiment, init_params[sl], max_copy_number, remap_cache=remap_cache,
                              seeds=(list(seeds)[sl] if seeds is not None else None), **kw)
        if groups > 1:
            # the groups are built side by side (a batch's construction is mostly device allocation and table building inside one C call,
            # which releases the GIL) once the segment remap they share is in the cache: one model built ahead of them fills it
            mk = dict((k, v) for k, v in kwargs.items() if k not in ('strict', 'mstep_threads', 'lockstep', 'native_search', 'sample_prep', 'options', 'h_init',
                                                                     'joint_accept', 'num_clones', 'device', 'quiet', 'kernel_module'))
            BreakpointModel(experiment.x, experiment.l, experiment.adjacencies, experiment.breakpoints, max_copy_number=max_copy_number,
                            divergence_weight=init_params[0]['divergence_weight'], max_depth=init_params[0]['max_depth'],
                            kernel_module=kwargs.get('kernel_module'), device=kwargs.get('device', 0), quiet=True, remap_cache=remap_cache, **mk)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=groups) as pool:
                self.sets = list(pool.map(build, self.slices))
        else:
            self.sets = [build(self.slices[0])]
        self.models = [m for rs in self.sets for m in rs.models]
        self.init_params = init_params
        self.experiment = experiment
        self._pool = None
        # paced: a group reaches each sweep's forward-backward point after its previous forward-backward launch has finished on the
        # device (library option pace_sweeps), instead of queueing all its sweeps at once; False = free-running.  Measured (DESIGN 4.6,
        # tools/s355_groups.sh): at 165 states free-running wins (407 it/s; paced 369: the groups already hide each one's marginal pass
        # under the other's forward-backward); at 355 states, where a forward-backward launch is 13 ms, free-running groups serialise
        # (118 it/s, one group of 16: 134) and pacing wins (144).  Groups of at most 4 restarts (a rank's share of 8 when 64 restarts are
        # sharded over 8 GPUs) also gain from pacing at 165 states.  'auto': pacing above 200 states or up to 4 restarts per group.
        if paced == 'auto':
            b0 = self.sets[0].batch
            small = max(len(rs.models) for rs in self.sets) <= 4
            paced = len(self.sets) >= 2 and b0 is not None and (getattr(b0, 'num_cn_states', 0) > 200 or small)
        native = all(getattr(rs.batch, 'set_option', None) is not None for rs in self.sets)
        self.paced = bool(paced) and len(self.sets) >= 2 and native
        if self.paced:
            for rs in self.sets:
                rs.batch.set_option('pace_sweeps', 1)
        # (Until round 4 this class also picked the parameter-search driver by grouping -- device-driven rounds for a single group and paced
        # groups, host-driven ones for free-running groups.  The device-driven rounds are the library's default for every batch now
        # (include/remixt_amd.h RMX_OPT_SEARCH_MODE): the search arithmetic no longer depends on the grouping.)

    def close(self):
        for rs in self.sets:
            rs.close()

    @property
    def num_restarts(self):
        return len(self.models)

    @property
    def batches(self):
        return [rs.batch for rs in self.sets]

    def _map(self, fn):
        if len(self.sets) == 1:
            return [fn(self.sets[0])]
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self.sets))
        return list(self._pool.map(fn, self.sets))

    def calculate_elbo(self):
        return np.concatenate(self._map(lambda rs: np.asarray(rs.calculate_elbo())))

    def variational_update(self, iters=1):
        self._map(lambda rs: rs.variational_update(iters))

    def em_iteration(self, i=0, num_update_iter=5):
        return np.concatenate(self._map(lambda rs: np.asarray(rs.em_iteration(i, num_update_iter))))

    def fit(self, num_em_iter=5, num_update_iter=5):
        return np.concatenate(self._map(lambda rs: np.asarray(rs.fit(num_em_iter, num_update_iter))))

    def run(self, num_em_iter, start=0, num_update_iter=5):
        """`num_em_iter` EM iterations of every restart.  The groups free-run (no join between
        iterations), which is what lets one group's sweeps overlap another group's M-step."""
        def go(rs):
            elbo = None
            for i in range(num_em_iter):
                elbo = rs.em_iteration(start + i, num_update_iter, defer_elbo=i + 1 < num_em_iter)
            return np.asarray(elbo)
        return np.concatenate(self._map(go))

    def synchronize(self):
        for rs in self.sets:
            if rs.batch is not None:
                rs.batch.synchronize()

    def results(self):
        return [r for part in self._map(lambda rs: rs.results()) for r in part]

    def profile(self):
        """{kernel: (ms, launches)} summed over the groups' batches."""
        out = {}
        for rs in self.sets:
            if rs.batch is None:
                continue
            for k, (ms, n) in rs.batch.profile().items():
                a = out.get(k, (0., 0))
                out[k] = (a[0] + ms, a[1] + n)
        return out


class DatasetGroups(object):
    """Several datasets resident on one GPU at once (BASELINE configs[4]: the tumour samples of one patient share
    the segmentation and the breakpoints and are fitted independently, reference remixt/workflow.py:472-485): one
    RestartGroups per dataset, free-running on their own host threads and HIP streams.  Nothing is shared
    between datasets on the device, so every dataset's results equal its single-dataset fit bit for bit.

    groups: restart groups PER DATASET; None (default) = as many as keep TWO restart groups on the device at a time -- two for one
    dataset, one each from two datasets on.  More than two groups at once share the runtime's four hardware queues (a group has two
    streams) and their kernels take turns: two datasets of 8 restarts measured 304-349 EM it/s as 2 x 2 groups against 449 as 2 x 1
    (profiles/r05_grouping_experiments.txt).  For the same reason at most `concurrent_groups` (default 2) groups run at once: three
    or more datasets are fitted two at a time, every one resident from the start."""

    def __init__(self, experiments, init_params, max_copy_number, groups=None, seeds=None, concurrent_groups=2, **kwargs):
        if len(experiments) != len(init_params):
            raise ValueError('one list of restarts per dataset')
        if groups is None:
            groups = max(1, int(concurrent_groups) // max(1, len(experiments)))
        self.groups_per_dataset = int(groups)
        self._workers = max(1, min(len(experiments), int(concurrent_groups) // max(1, self.groups_per_dataset)))
        self.parts = [RestartGroups(e, p, max_copy_number, groups=groups, seeds=(seeds[i] if seeds is not None else None), **kwargs)
                      for i, (e, p) in enumerate(zip(experiments, init_params))]
        self.sets = [rs for part in self.parts for rs in part.sets]
        self.models = [m for part in self.parts for m in part.models]
        self.init_params = [p for part in self.parts for p in part.init_params]
        self.experiments = list(experiments)
        self._pool = None

    def close(self):
        for rs in self.sets:
            rs.close()

    @property
    def num_restarts(self):
        return len(self.models)

    @property
    def batches(self):
        return [rs.batch for rs in self.sets]

    def _map(self, fn):
        if len(self.parts) == 1:
            return [fn(self.parts[0])]
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self._workers)
        return list(self._pool.map(fn, self.parts))

    def calculate_elbo(self):
        return np.concatenate(self._map(lambda part: part.calculate_elbo()))

    def variational_update(self, iters=1):
        self._map(lambda part: part.variational_update(iters))

    def fit(self, num_em_iter=5, num_update_iter=5):
        return np.concatenate(self._map(lambda part: part.fit(num_em_iter, num_update_iter)))

    def run(self, num_em_iter, start=0, num_update_iter=5):
        return np.concatenate(self._map(lambda part: part.run(num_em_iter, start, num_update_iter)))

    def synchronize(self):
        for part in self.parts:
            part.synchronize()

    def results(self):
        """Per-restart result dicts, dataset after dataset (each with 'dataset' = its index)."""
        out = []
        for i, part in enumerate(self._map(lambda part: part.results())):
            for r in part:
                r['dataset'] = i
                out.append(r)
        return out

    def results_by_dataset(self):
        return self._map(lambda part: part.results())

    def profile(self):
        out = {}
        for part in self.parts:
            for k, (ms, n) in part.profile().items():
                a = out.get(k, (0., 0))
                out[k] = (a[0] + ms, a[1] + n)
        return out


def collect_fit_results(model, experiment, init_params, cn=None, indicators=None):
    """fit_results of analysis/pipeline.py:196-226 from a fitted BreakpointModel (`cn`: its Viterbi
    path if a batched decode already produced it; `indicators`: its (p_outlier_total, p_outlier_allele) in model
    segment order if a batched transfer already fetched them)."""
    from .cn_model import decode_breakpoints_naive
    cn, brk_cn = model.optimal_cn(cn) if cn is not None else model.optimal_cn()
    if model.disable_breakpoints:
        brk_cn = decode_breakpoints_naive(cn, experiment.adjacencies, experiment.breakpoints)
    res = dict()
    res['h'] = np.array(model.h, dtype=float)       # a copy: a kernel model may hand out a view of its own buffer
    res['cn'] = cn
    res['brk_cn'] = brk_cn
    if indicators is not None:
        res['p_outlier_total'] = np.asarray(indicators[0])[model.seg_fwd_remap]          # (fancy indexing: a copy, as the properties return)
        res['p_outlier_allele'] = np.asarray(indicators[1])[model.seg_fwd_remap]
    else:
        res['p_outlier_total'] = np.array(model.p_outlier_total)
        res['p_outlier_allele'] = np.array(model.p_outlier_allele)
    res['total_likelihood_mask'] = np.array(model.total_likelihood_mask)
    res['allele_likelihood_mask'] = np.array(model.allele_likelihood_mask)
    stats = dict()
    stats['elbo'] = model.prev_elbo
    stats['elbo_diff'] = model.prev_elbo_diff
    stats['error_message'] = ''
    stats.update(model.get_likelihood_param_values())
    l = np.asarray(experiment.l)
    # analysis/pipeline.py:215-217: ploidy = (cn[:,1:,:].mean(axis=1).T * l).sum() / l.sum(), divergent = (max over the tumour clones != min).
    # Reductions over an axis of two or three elements are numpy's slow case (23 ms of a 9 ms-per-restart budget at 50 000 segments): the
    # same operations clone by clone -- the same floating-point values in the same order (tests/test_host_golden.py compares the bits)
    ploidy, divergent = tumour_ploidy_and_divergence(cn, l)
    stats['num_clones'] = len(model.h)
    stats['num_segments'] = len(experiment.x)
    stats['ploidy'] = ploidy
    stats['proportion_divergent'] = (divergent.T * l).sum() / (2. * l.sum())
    stats['mode_idx'] = init_params.get('mode_idx', 0)
    stats['divergence_weight'] = init_params['divergence_weight']
    res['stats'] = stats
    return res


def tumour_ploidy_and_divergence(cn, l):
    """(length-weighted mean tumour copies per allele, per-segment-and-allele indicator of clones that differ) of a decoded copy number
    cn (N, M, 2) -- the two statistics of analysis/pipeline.py:215-217, bit for bit."""
    t = np.asarray(cn)[:, 1:, :]
    acc = t[:, 0, :].astype(float)
    lo = t[:, 0, :].copy(); hi = t[:, 0, :].copy()
    for m in range(1, t.shape[1]):
        acc = acc + t[:, m, :]
        np.minimum(lo, t[:, m, :], out=lo); np.maximum(hi, t[:, m, :], out=hi)
    mean = acc / t.shape[1]
    return (mean.T * l).sum() / l.sum(), (hi != lo) * 1.


# ---------------------------------------------------------------------------------
# multi-GPU: shard restarts, gather results
# ---------------------------------------------------------------------------------
def shard_indices(num_items, world_size, rank):
    """Restart i -> rank i mod world_size."""
    return list(range(rank, num_items, world_size))


_HDR = 5      # elbo, elbo_diff, ploidy, proportion_divergent, failure code
_FAILURE_TEXT = {1: 'optimization failed (h kept)', 2: 'gradiant error (h kept)', 3: 'restart failed'}


def _failure_code(message):
    """A restart's error message as a number for the fixed-size float record (0 = none)."""
    if not message:
        return 0
    for code, text in _FAILURE_TEXT.items():
        if message.startswith(text.split(' (')[0]):
            return code
    return 3


def _pack(res, N, M, K, nparams, brk_ids, param_names):
    """One restart's results as (float64 vector, int8 vector) of fixed length."""
    f = np.zeros(_HDR + M + nparams + 4 * N, dtype=np.float64)
    st = res['stats']
    f[0] = st['elbo']; f[1] = st['elbo_diff'] if st['elbo_diff'] is not None else np.nan
    f[2] = st['ploidy']; f[3] = st['proportion_divergent']
    f[4] = float(_failure_code(st.get('error_message', '')))
    f[_HDR:_HDR + M] = res['h']
    f[_HDR + M:_HDR + M + nparams] = [st[k] for k in param_names]
    o = _HDR + M + nparams
    f[o:o + 2 * N] = res['p_outlier_total'].ravel(); f[o + 2 * N:o + 4 * N] = res['p_outlier_allele'].ravel()
    i8 = np.zeros(N * M * 2 + K * M + 2 * N, dtype=np.int8)
    i8[:N * M * 2] = res['cn'].ravel()
    i8[N * M * 2:N * M * 2 + K * M] = np.array([res['brk_cn'][k] for k in brk_ids]).ravel()
    i8[N * M * 2 + K * M:N * M * 2 + K * M + N] = res['total_likelihood_mask']
    i8[N * M * 2 + K * M + N:] = res['allele_likelihood_mask']
    return f, i8


def _unpack(f, i8, N, M, K, nparams, brk_ids, param_names, init_params):
    res = dict()
    res['h'] = f[_HDR:_HDR + M].copy()
    o = _HDR + M + nparams
    res['p_outlier_total'] = f[o:o + 2 * N].reshape(N, 2).copy()
    res['p_outlier_allele'] = f[o + 2 * N:o + 4 * N].reshape(N, 2).copy()
    res['cn'] = i8[:N * M * 2].astype(np.int64).reshape(N, M, 2)
    bc = i8[N * M * 2:N * M * 2 + K * M].astype(np.int64).reshape(K, M)
    res['brk_cn'] = dict((k, bc[i]) for i, k in enumerate(brk_ids))
    res['total_likelihood_mask'] = i8[N * M * 2 + K * M:N * M * 2 + K * M + N].astype(np.int64)
    res['allele_likelihood_mask'] = i8[N * M * 2 + K * M + N:].astype(np.int64)
    code = int(f[4]) if np.isfinite(f[4]) else 3
    st = {'elbo': float(f[0]), 'elbo_diff': float(f[1]), 'ploidy': float(f[2]), 'proportion_divergent': float(f[3]),
          'error_message': _FAILURE_TEXT.get(code, '') if code else '', 'num_clones': M, 'num_segments': N,
          'mode_idx': init_params.get('mode_idx', 0), 'divergence_weight': init_params['divergence_weight']}
    for j, k in enumerate(param_names):
        st[k] = float(f[_HDR + M + j])
    res['stats'] = st
    return res


def fit_restarts_distributed(experiment, init_params, max_copy_number, num_clones=3, num_em_iter=5, num_update_iter=5,
                             device=None, kernel_module=None, seeds=None, quiet=True, groups=2, **model_kwargs):
    """Fit all restarts across the ranks of the default torch.distributed group.

    Every rank holds the (small, read-only) experiment; rank g fits restarts
    g, g+G, g+2G, ... on its own GPU; one all-gather of fixed-size result records
    returns every restart's results to every rank (`collate` stores all of them,
    analysis/pipeline.py:289-291).  Works on one process without torch.distributed.
    """
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    mine = shard_indices(len(init_params), world, rank)
    if device is None:
        device = (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    local = []
    param_names = None
    if mine:
        my_seeds = [seeds[i] for i in mine] if seeds is not None else None
        rs = RestartGroups(experiment, [init_params[i] for i in mine], max_copy_number, groups=(groups if my_seeds is not None else 1),
                           num_clones=num_clones, device=device, quiet=quiet, kernel_module=kernel_module, seeds=my_seeds,
                           **model_kwargs)
        rs.fit(num_em_iter, num_update_iter)
        local = rs.results()
        param_names = list(rs.models[0].likelihood_params)
        rs.close()      # (the batches' device memory and streams now, not when the collector gets to them: DESIGN 4.6)
    if param_names is None:
        nc = model_kwargs.get('normal_contamination', True)
        param_names = ['negbin_r_0', 'negbin_r_1', 'betabin_M_0', 'betabin_M_1'] + (
            [] if nc else ['negbin_hdel_mu', 'negbin_hdel_r_0', 'negbin_hdel_r_1', 'betabin_loh_p', 'betabin_loh_M_0', 'betabin_loh_M_1'])
    return gather_result_records(local, experiment, init_params, num_clones, param_names, device=device)


def gather_result_records(local, experiment, init_params, num_clones, param_names, device=None, timing=None, local_ids=None):
    """The one collective of the path: every rank contributes the fixed-size records of the restarts
    it fitted (`local`, in the order of shard_indices) -- one float64 record (ELBO, h, parameters, outlier
    probabilities, failure code) and one int8 record (cn, brk_cn, masks) per restart -- and every rank gets the
    results of all `len(init_params)` restarts back, keyed by restart id (`collate` stores all of them,
    analysis/pipeline.py:289-291).  Two all_gathers (RCCL over xGMI with the nccl backend; gloo in the CPU tests); a
    process without a process group just repacks.  `timing`: a dict that receives the message size and the seconds
    spent in the collectives.  `local_ids`: the restart ids of `local` when the ranks' shares are not shard_indices'
    (several datasets cut over the ranks as one list of units); they are gathered first (one small all_gather)."""
    import time
    import torch
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if distributed else 1
    N = len(experiment.x); M = num_clones
    brk_ids = list(experiment.breakpoints.keys()); K = len(brk_ids)
    nparams = len(param_names)
    per_rank = (len(init_params) + world - 1) // world
    if local_ids is not None:
        # shares that are not shard_indices' (units of several datasets cut over the ranks) need not be balanced per dataset: every rank's
        # buffer takes the largest share
        per_rank = max(per_rank, len(local_ids), 1)
        if distributed:
            dev_ = torch.device('cuda', device if device is not None else torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
            cnt_ = torch.tensor([per_rank], dtype=torch.int64, device=dev_)
            dist.all_reduce(cnt_, op=dist.ReduceOp.MAX)
            per_rank = int(cnt_.item())
    if len(local) > per_rank:
        raise ValueError('gather_result_records: %d local results for a share of %d (pass local_ids for shares that are not shard_indices\')' % (len(local), per_rank))
    flen = _HDR + M + nparams + 4 * N
    ilen = N * M * 2 + K * M + 2 * N
    fbuf = np.full((per_rank, flen), np.nan); ibuf = np.zeros((per_rank, ilen), dtype=np.int8)
    for j, res in enumerate(local):
        fbuf[j], ibuf[j] = _pack(res, N, M, K, nparams, brk_ids, param_names)
    if local_ids is not None:
        ids_t = np.full((per_rank,), -1, dtype=np.int64); ids_t[:len(local_ids)] = local_ids
    t0 = time.perf_counter()
    ids_all = None
    if distributed and local_ids is not None:
        dev_ = torch.device('cuda', device if device is not None else torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
        it_ = torch.from_numpy(ids_t).to(dev_)
        got = [torch.empty_like(it_) for _ in range(world)]
        dist.all_gather(got, it_)
        ids_all = [t.cpu().numpy() for t in got]
    elif local_ids is not None:
        ids_all = [ids_t]
    if distributed:
        on_gpu = dist.get_backend() == 'nccl'
        if device is None:
            device = torch.cuda.current_device() if on_gpu else 0
        dev = torch.device('cuda', device) if on_gpu else torch.device('cpu')
        ft = torch.from_numpy(fbuf).to(dev); it = torch.from_numpy(ibuf).to(dev)
        fall = [torch.empty_like(ft) for _ in range(world)]; iall = [torch.empty_like(it) for _ in range(world)]
        dist.all_gather(fall, ft)     # RCCL over xGMI on the GPU box; gloo in CPU tests
        dist.all_gather(iall, it)
        fall = [t.cpu().numpy() for t in fall]; iall = [t.cpu().numpy() for t in iall]
    else:
        fall, iall = [fbuf], [ibuf]
    if timing is not None:
        timing.update({'seconds': time.perf_counter() - t0, 'bytes_per_rank': int(fbuf.nbytes + ibuf.nbytes),
                       'float_record_bytes': int(flen * 8), 'int8_record_bytes': int(ilen), 'records_per_rank': int(per_rank)})
    results = {}
    for g in range(world):
        ids_g = shard_indices(len(init_params), world, g) if ids_all is None else [int(i) for i in ids_all[g] if i >= 0]
        for j, i in enumerate(ids_g):
            results[i] = _unpack(fall[g][j], iall[g][j], N, M, K, nparams, brk_ids, param_names, init_params[i])
    return results


def select_optimal(results, max_prop_diverge=0.5):
    """store_optimal_solution (analysis/pipeline.py:253-264): best ELBO among solutions
    with proportion_divergent < max_prop_diverge (all solutions if none qualifies).

    In the reference a restart whose h M-step fails raises (cn_model.py:510-521) and takes the whole
    workflow down; here such a restart is recorded (stats['error_message']) and can never be selected.
    A NaN ELBO sorts last, as in pandas' sort_values(ascending=False)."""
    ids = sorted(results)
    alive = [i for i in ids if not results[i]['stats'].get('error_message')]
    if not alive:
        raise ValueError('every restart failed: ' + '; '.join(sorted(set(results[i]['stats']['error_message'] for i in ids))))
    ok = [i for i in alive if results[i]['stats']['proportion_divergent'] < max_prop_diverge]
    pool = ok if ok else alive
    finite = [i for i in pool if np.isfinite(results[i]['stats']['elbo'])]
    pool = finite if finite else pool
    # pandas sort_values(ascending=False) is a stable sort on -elbo: first maximum wins
    best = max(pool, key=lambda i: (results[i]['stats']['elbo'], -i))
    return best
