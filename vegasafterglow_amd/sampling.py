"""Sampler-side glue for the batched device likelihood.

The reference drives emcee with ``vectorize=True`` and a ``log_prob_batch(samples[nb, ndim])`` closure that maps
walkers over a thread pool (VegasAfterglow/fitting/samplers.py:59-106).  Here the closure is one device call
(``Fitter.make_log_prob_batch``), so a sampler only has to hand over whole walker blocks:

* ``emcee_sampler``  -- the reference's own sampler object, wired to the device closure (needs emcee installed);
* ``run_stretch_move`` -- a dependency-free affine-invariant ensemble sampler (Goodman & Weare stretch move with the
  red-blue split emcee uses by default), so an end-to-end fit runs wherever the engine does.  Each half-step
  proposes nwalkers/2 walkers and evaluates them in ONE ``log_prob_batch`` call;
* ``BatchPool`` -- the object handed to ``bilby.run_sampler(pool=...)`` in place of the reference's thread pool
  (samplers.py:146-170): ``pool.map(loglikelihood, points)`` becomes one device call over the whole live-point queue.
"""
from typing import Callable, Optional, Sequence

import numpy as np

from .fitting import Fitter, ParamDef


def initial_positions(lower: np.ndarray, upper: np.ndarray, nwalkers: int, rng: np.random.Generator,
                      center: Optional[np.ndarray] = None, spread: float = 0.1) -> np.ndarray:
    """Walkers in a ball of relative width `spread` (of the box) around `center` (default: the box centre), clipped
    to the bounds -- the strategy of generate_initial_positions (fitting/samplers.py) for uniform priors."""
    lower, upper = np.asarray(lower, dtype=np.float64), np.asarray(upper, dtype=np.float64)
    c = 0.5 * (lower + upper) if center is None else np.asarray(center, dtype=np.float64)
    pos = c + spread * (upper - lower) * rng.standard_normal((nwalkers, lower.size))
    eps = 1e-9 * (upper - lower)
    return np.clip(pos, lower + eps, upper - eps)


def emcee_sampler(fitter: Fitter, param_defs: Sequence[ParamDef], nwalkers: int, loglike_fn: Optional[Callable] = None,
                  moves=None):
    """emcee.EnsembleSampler(nwalkers, ndim, log_prob_batch, vectorize=True, moves=moves) exactly like
    fitting/samplers.py:109-115, with the device closure in place of the thread-pool one."""
    import emcee  # optional dependency, as in the reference

    log_prob_batch = fitter.make_log_prob_batch(param_defs, loglike_fn=loglike_fn)
    spec, _, _ = fitter.build_spec(param_defs)
    return emcee.EnsembleSampler(nwalkers, spec.ndim, log_prob_batch, vectorize=True, moves=moves)


class BatchPool:
    """Drop-in for ThreadPoolWithClose (fitting/samplers.py:146-170) as bilby / dynesty use it: with
    ``use_pool={"loglikelihood": True}`` the sampler calls ``pool.map(loglikelihood, queue)`` with ``queue_size`` points
    in sampler space; here the whole queue goes through ``batch_fn(points[n, ndim]) -> float64[n]`` (one device call,
    e.g. ``fitter.make_log_prob_batch(defs)`` or ``lambda v: fitter.loglike_batch(v, defs)``) and the per-point callable
    is only used for anything that is not a point queue."""

    def __init__(self, batch_fn: Callable[[np.ndarray], np.ndarray], ndim: int, size: int = 1024):
        self.batch_fn, self.ndim, self.size = batch_fn, int(ndim), int(size)  # `size`: queue length dynesty should fill
        self.calls = 0

    def map(self, func, iterable, chunksize=None):
        tasks = list(iterable)
        if not tasks:
            return []
        try:
            pts = np.asarray(tasks, dtype=np.float64)
        except (TypeError, ValueError):
            pts = None
        if pts is None or pts.ndim != 2 or pts.shape[1] != self.ndim:
            return [func(x) for x in tasks]
        self.calls += 1
        out = np.asarray(self.batch_fn(pts), dtype=np.float64)
        out[~np.isfinite(out)] = -np.inf  # non-finite likelihoods are rejected points (samplers.py:66-70)
        return list(out)

    def close(self):
        pass

    def join(self):
        pass

    def shutdown(self, wait=True):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def run_stretch_move(log_prob_batch: Callable[[np.ndarray], np.ndarray], pos0: np.ndarray, nsteps: int,
                     rng: Optional[np.random.Generator] = None, a: float = 2.0, progress: Optional[Callable] = None):
    """Affine-invariant ensemble MCMC (Goodman & Weare 2010, stretch move, red-blue halves).

    Returns (chain[nsteps, nwalkers, ndim], log_prob[nsteps, nwalkers], acceptance_fraction[nwalkers]).
    `log_prob_batch` is called twice per step with nwalkers/2 rows each."""
    rng = np.random.default_rng() if rng is None else rng
    pos = np.array(pos0, dtype=np.float64, copy=True)
    nwalkers, ndim = pos.shape
    if nwalkers < 2 * ndim or nwalkers % 2:
        raise ValueError("need an even number of walkers, at least 2 * ndim")
    lp = np.asarray(log_prob_batch(pos), dtype=np.float64)
    if not np.all(np.isfinite(lp)):
        raise ValueError("initial positions must have finite log-probability")
    chain = np.empty((nsteps, nwalkers, ndim))
    logp = np.empty((nsteps, nwalkers))
    accepted = np.zeros(nwalkers)
    half = nwalkers // 2
    idx = np.arange(nwalkers)
    for step in range(nsteps):
        perm = rng.permutation(nwalkers)
        for first in (True, False):
            s = perm[:half] if first else perm[half:]  # walkers being updated
            c = perm[half:] if first else perm[:half]  # the complementary ensemble
            z = ((a - 1.0) * rng.random(half) + 1.0) ** 2 / a  # g(z) ~ 1/sqrt(z) on [1/a, a]
            partner = pos[c[rng.integers(half, size=half)]]
            prop = partner + z[:, None] * (pos[s] - partner)
            lp_new = np.asarray(log_prob_batch(prop), dtype=np.float64)
            lp_new[~np.isfinite(lp_new)] = -np.inf
            log_ratio = (ndim - 1.0) * np.log(z) + lp_new - lp[s]
            take = np.log(rng.random(half)) < log_ratio
            pos[s[take]] = prop[take]
            lp[s[take]] = lp_new[take]
            accepted[s[take]] += 1
        chain[step], logp[step] = pos, lp
        if progress is not None:
            progress(step, pos, lp)
    del idx
    return chain, logp, accepted / max(nsteps, 1)


def fit(fitter: Fitter, param_defs: Sequence[ParamDef], nwalkers: int, nsteps: int, nburn: int = 0, seed: int = 0,
        loglike_fn: Optional[Callable] = None, center: Optional[np.ndarray] = None, spread: float = 0.1):
    """Convenience driver: uniform priors over the ParamDef boxes, stretch-move sampling on the device likelihood.
    Returns dict(samples[flat, ndim], log_prob[flat], acceptance, best) in sampler space (log10 for LOG parameters)."""
    rng = np.random.default_rng(seed)
    _, lower, upper = fitter.build_spec(param_defs)
    log_prob_batch = fitter.make_log_prob_batch(param_defs, loglike_fn=loglike_fn)
    pos0 = initial_positions(lower, upper, nwalkers, rng, center=center, spread=spread)
    chain, logp, acc = run_stretch_move(log_prob_batch, pos0, nsteps, rng=rng)
    flat = chain[nburn:].reshape(-1, chain.shape[-1])
    flat_lp = logp[nburn:].reshape(-1)
    return {"samples": flat, "log_prob": flat_lp, "acceptance": acc, "best": flat[np.argmax(flat_lp)], "chain": chain}
