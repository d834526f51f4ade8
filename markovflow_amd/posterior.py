"""
Posterior processes: prediction at new time points from the smoothed chain.

Mirror of ``ConditionalProcess`` / ``AnalyticPosteriorProcess`` of ``markovflow/posterior.py:160-258,416-470`` (reference):
``predict_state`` conditions every new time point on the pairwise posterior marginal of its two neighbouring training
points (``markovflow/conditionals.py:29-83,122-256,380-485``).  The neighbour search is ``torch.searchsorted`` (the reference
uses ``tf.searchsorted``); the transitions to / from the new points come from ``mf_sde_matern_transitions_*`` and the
conditional statistics + projection + marginalisation run in ONE HIP kernel (``mf_sde_conditional_predict_*``, a lane per
new point).  ``sample_state_trajectories`` / ``sample_state`` / ``sample_f`` (posterior.py:45-138,260-412) draw joint samples by
Matheron's rule, as the reference does: a joint prior sample at the new and the conditioning points (the kernel's state space
model on the merged, sorted time points - generated and propagated by the HIP kernels), a posterior sample at the conditioning
points, and the local conditional-mean correction.  Mean functions are not mirrored (zero mean).
"""
from typing import Optional, Sequence, Tuple, Union

import torch

from . import _lib
from .gauss_markov import GaussMarkovDistribution
from .kernels import SDEKernel

APPROX_INF = 1e10   # markovflow/base.py:46: the "time" of the stationary prior beyond both ends of the data


class ConditionalProcess:
    """Posterior process built from the marginals ``q(s(Z))`` and the prior conditional ``p(s(.)|s(Z))`` (posterior.py:160-258)."""

    def __init__(self, posterior_dist: GaussMarkovDistribution, kernel: SDEKernel, conditioning_time_points: torch.Tensor) -> None:
        self.gauss_markov_model = posterior_dist
        self.kernel = kernel
        self.conditioning_time_points = conditioning_time_points

    def predict_state(self, new_time_points: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """
        State mean ``batch + [num_new, state_dim]`` and covariance ``batch + [num_new, state_dim, state_dim]`` at
        ``new_time_points`` (``batch + [num_new]``, sorted) - posterior.py:207-229.
        """
        dist, kern = self.gauss_markov_model, self.kernel
        train = self.conditioning_time_points
        batch = tuple(dist.batch_shape)
        if tuple(new_time_points.shape[:-1]) != batch or tuple(train.shape[:-1]) != batch:
            raise ValueError("new_time_points and conditioning_time_points must carry the distribution's batch shape")
        n, n_new, d = train.shape[-1], new_time_points.shape[-1], dist.state_dim
        dtype, dev = train.dtype, train.device
        new = new_time_points.to(dtype).contiguous()
        # insertion indices and the gaps to the neighbours (conditionals.py:231-245; the prior sits at -/+ APPROX_INF)
        idx = torch.searchsorted(train.contiguous(), new)
        inf = torch.full(batch + (1,), APPROX_INF, dtype=dtype, device=dev)
        aug = torch.cat([-inf, train, inf], dim=-1)
        minus, plus = torch.gather(aug, -1, idx), torch.gather(aug, -1, idx + 1)
        m0 = kern.initial_mean(batch).to(dtype=dtype, device=dev).expand(batch + (d,)).contiguous()
        p0 = kern.initial_covariance(new[..., :1]).to(dtype=dtype, device=dev)
        p0 = p0.expand(batch + (d, d)).contiguous()
        if d > _lib.load().mf_max_state_dim():
            # beyond the lane-per-point kernel (d <= 9): the reference's own composition (posterior.py:217-221) of the functions of
            # conditionals.py - pairwise_marginals -> conditional_predict - as batched products (the posterior chain and its moments
            # come from the row kernels / the tile engine).  Taken BEFORE the transitions and moments below are formed: those
            # functions form their own
            from . import conditionals
            pair_mean, pair_cov = conditionals.pairwise_marginals(dist, m0, p0)
            return conditionals.conditional_predict(new, train, kern, pair_mean, pair_cov)
        a_mt, q_mt = kern.transition_statistics(minus, new - minus)
        a_tp, q_tp = kern.transition_statistics(new, plus - new)
        means, covs, sub = dist._moments(want_sub=n > 1)
        flat = lambda t, k: t.reshape((-1,) + tuple(t.shape[-k:])).contiguous()  # noqa: E731
        bsz = max(1, int(torch.tensor(batch).prod())) if batch else 1
        out_mean = torch.empty((bsz, n_new, d), dtype=dtype, device=dev)
        out_cov = torch.empty((bsz, n_new, d, d), dtype=dtype, device=dev)
        info = _lib.pivot_info(dev)
        _lib.call("mf_sde_conditional_predict", dtype, bsz, n, n_new, d, _lib.ptr(flat(idx, 1)), _lib.ptr(flat(a_mt, 3)),
                  _lib.ptr(flat(q_mt, 3)), _lib.ptr(flat(a_tp, 3)), _lib.ptr(flat(q_tp, 3)), _lib.ptr(flat(means, 2)),
                  _lib.ptr(flat(covs, 3)), _lib.ptr(None if sub is None else flat(sub, 3)), _lib.ptr(flat(m0, 1)),
                  _lib.ptr(flat(p0, 2)), _lib.ptr(out_mean), _lib.ptr(out_cov), info, _lib.stream_ptr(dev))
        _lib.raise_on_info(info, "ConditionalProcess.predict_state", dev)
        return out_mean.reshape(batch + (n_new, d)), out_cov.reshape(batch + (n_new, d, d))

    def sample_state_trajectories(self, new_time_points: torch.Tensor, sample_shape: Union[int, Sequence[int]],
                                  *, input_data=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Joint state samples at ``new_time_points`` and at the conditioning points (posterior.py:260-374; Appendix 2 of
        "Doubly Sparse Variational Gaussian Processes"):  ``[s_p, u_p] ~ p(s([t, z]))``, ``u_o ~ q(s(z))``,
        ``s_o = s_p - E[s(t) | s(z) = u_p - u_o]`` - the conditional mean only involves the two conditioning states around each
        new point.  Returns ``(sample_shape + batch + [num_new, d], sample_shape + batch + [num_conditioning, d])``."""
        shape = (sample_shape,) if isinstance(sample_shape, int) else tuple(sample_shape)
        dist, kern, train = self.gauss_markov_model, self.kernel, self.conditioning_time_points
        batch = tuple(dist.batch_shape)
        new = new_time_points.to(train.dtype)
        n, n_new = train.shape[-1], new.shape[-1]
        joint = torch.cat([train, new], dim=-1)
        sort_ind = torch.argsort(joint, dim=-1)
        sorted_samples = kern.state_space_model(torch.gather(joint, -1, sort_ind).contiguous()).sample(shape)
        unsort = torch.argsort(sort_ind, dim=-1)
        unsort = unsort.expand(shape + tuple(unsort.shape))[..., None].expand(shape + batch + (n + n_new, sorted_samples.shape[-1]))
        joint_samples = torch.gather(sorted_samples, -2, unsort)
        prior_cond, prior_new = joint_samples[..., :n, :], joint_samples[..., n:, :]
        post_cond = dist.sample(shape)
        delta = prior_cond - post_cond
        # infinitely far from the data the posterior reverts to the prior: the correction vanishes beyond both ends
        pad = torch.zeros_like(delta[..., :1, :])
        delta_aug = torch.cat([pad, delta, pad], dim=-2)
        idx = torch.searchsorted(train.contiguous(), new.contiguous())
        gidx = idx.expand(shape + tuple(idx.shape))[..., None].expand(shape + batch + (n_new, delta.shape[-1]))
        u_minus, u_plus = torch.gather(delta_aug, -2, gidx), torch.gather(delta_aug, -2, gidx + 1)
        d_m, e_m = self._conditional_mean_projection(new, idx)
        correction = (d_m @ u_minus[..., None] + e_m @ u_plus[..., None])[..., 0]
        return prior_new - correction, post_cond

    def _conditional_mean_projection(self, new: torch.Tensor, idx: torch.Tensor):
        """``E[s(t) | s(z_-), s(z_+)] = D s(z_-) + E s(z_+)`` for every new point (conditionals.py:29-83,122-203):
        ``E = Q_mt A_tp^T (Q_tp + A_tp Q_mt A_tp^T)^-1``, ``D = A_mt - E A_tp A_mt`` from the prior transitions z_- -> t -> z_+."""
        kern, train = self.kernel, self.conditioning_time_points
        batch = tuple(train.shape[:-1])
        inf = torch.full(batch + (1,), APPROX_INF, dtype=train.dtype, device=train.device)
        aug = torch.cat([-inf, train, inf], dim=-1)
        minus, plus = torch.gather(aug, -1, idx), torch.gather(aug, -1, idx + 1)
        a_mt, q_mt = kern.transition_statistics(minus, new - minus)
        a_tp, q_tp = kern.transition_statistics(new, plus - new)
        g = a_tp @ q_mt
        e_m = torch.linalg.solve(q_tp + g @ a_tp.transpose(-1, -2), g).transpose(-1, -2)
        return a_mt - e_m @ a_tp @ a_mt, e_m

    def sample_state(self, new_time_points: torch.Tensor, sample_shape, *, input_data=None) -> torch.Tensor:
        """State samples at ``new_time_points``, ``sample_shape + batch + [num_new, d]`` (posterior.py:45-76)."""
        return self.sample_state_trajectories(new_time_points, sample_shape, input_data=input_data)[0]

    def sample_f(self, new_time_points: torch.Tensor, sample_shape, *, input_data=None) -> torch.Tensor:
        """Function samples (projected states), ``sample_shape + batch + [num_new, output_dim]`` (posterior.py:376-412; zero
        mean function)."""
        states = self.sample_state(new_time_points, sample_shape)
        return self.kernel.generate_emission_model(new_time_points).project_state_to_f(states)

    def predict_f(self, new_time_points: torch.Tensor, full_output_cov: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """Marginal function values at ``new_time_points``: means ``batch + [num_new, output_dim]`` and variances (or full
        output covariances) - posterior.py:231-258 (zero mean function)."""
        emission = self.kernel.generate_emission_model(new_time_points)
        return emission.project_state_marginals_to_f(*self.predict_state(new_time_points), full_output_cov=full_output_cov)


def _predict_state_dense(idx, a_mt, q_mt, a_tp, q_tp, means, covs, sub, m0, p0):
    """``mf_sde_conditional_predict_*`` (csrc/mf_kernels.hpp: sde_predict_kernel) as batched products, for state dimensions beyond
    that kernel: ``p(x_t) = N(D mu_- + E mu_+, T + [D E] S [D E]^T)`` with ``Q-+ = Q_tp + A_tp Q_mt A_tp^T``,
    ``E = Q_mt A_tp^T Q-+^-1``, ``D = A_mt - E A_tp A_mt``, ``T = Q_mt - Q_mt A_tp^T Q-+^-1 A_tp Q_mt`` and ``(mu_-, mu_+, S)`` the
    pairwise posterior marginal of the training points around ``t`` - the prior beyond the ends (conditionals.py:29-83,122-203,
    380-485).  ``idx [B, Np]``: insertion index of every new point; everything else flat over the batch."""
    tr = lambda t: t.transpose(-1, -2)                                   # noqa: E731
    tri = torch.linalg.solve_triangular
    n = means.shape[1]
    g = a_tp @ q_mt
    chol = _lib.checked_cholesky(q_tp + g @ tr(a_tp), "ConditionalProcess.predict_state")
    v = tri(chol, g, upper=False)                                         # L^-1 A_tp Q_mt
    t_m = q_mt - tr(v) @ v
    e = tr(tri(tr(chol), v, upper=True))                                  # E = (L^-T V)^T
    d_m = a_mt - e @ a_tp @ a_mt
    has_m, has_p = (idx > 0), (idx < n)
    pick = lambda src, i, k: torch.gather(src, 1, i.reshape(i.shape + (1,) * k).expand(i.shape + tuple(src.shape[2:])))  # noqa: E731
    im, ip = (idx - 1).clamp(min=0), idx.clamp(max=n - 1)
    mu_m = torch.where(has_m[..., None], pick(means, im, 1), m0[:, None, :])
    mu_p = torch.where(has_p[..., None], pick(means, ip, 1), m0[:, None, :])
    mean = (d_m @ mu_m[..., None] + e @ mu_p[..., None])[..., 0]
    p_m = torch.where(has_m[..., None, None], pick(covs, im, 2), p0[:, None])
    p_p = torch.where(has_p[..., None, None], pick(covs, ip, 2), p0[:, None])
    x1, x2 = d_m @ p_m, e @ p_p
    if sub is not None and n > 1:
        c = pick(sub, im.clamp(max=n - 2), 2) * (has_m & has_p)[..., None, None].to(sub.dtype)     # Cov(x_+, x_-)
        x1, x2 = x1 + e @ c, x2 + d_m @ tr(c)
    return mean, t_m + x1 @ tr(d_m) + x2 @ tr(e)


class AnalyticPosteriorProcess(ConditionalProcess):
    """Posterior process of a model with an analytic (Gaussian) likelihood: adds ``predict_y`` (posterior.py:416-470)."""

    def __init__(self, posterior_dist: GaussMarkovDistribution, kernel: SDEKernel, conditioning_time_points: torch.Tensor,
                 chol_obs_covariance: Optional[torch.Tensor] = None) -> None:
        super().__init__(posterior_dist, kernel, conditioning_time_points)
        self._chol_obs_covariance = chol_obs_covariance

    def predict_y(self, new_time_points: torch.Tensor, full_output_cov: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """Observation marginals: ``predict_f`` plus the noise covariance (posterior.py:445-470)."""
        f_mean, f_cov = self.predict_f(new_time_points, full_output_cov=full_output_cov)
        if self._chol_obs_covariance is None:
            return f_mean, f_cov
        noise = self._chol_obs_covariance @ self._chol_obs_covariance.transpose(-1, -2)
        return f_mean, f_cov + (noise if full_output_cov else torch.diagonal(noise, dim1=-2, dim2=-1))
