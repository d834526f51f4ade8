"""
Thin model harness over the Kalman path: ``GaussianProcessRegression`` (mirror of
``markovflow/models/gaussian_process_regression.py:29-160``, the direct caller of ``KalmanFilter`` in the reference).
Only what drives the hot path is mirrored: construction from ``(time_points, observations)`` and an SDE kernel,
``log_likelihood`` / ``loss`` and the posterior state space model; prediction at new time points
(``AnalyticPosteriorProcess``), mean functions and training loops belong to the reference's outer layers.
"""
from typing import Optional, Tuple

import torch

from .kalman_filter import KalmanFilter
from .kernels import SDEKernel
from .state_space_model import StateSpaceModel


class GaussianProcessRegression:
    """GP regression as a Kalman filter on the kernel's state space model (gaussian_process_regression.py:29-160)."""

    def __init__(self, input_data: Tuple[torch.Tensor, torch.Tensor], kernel: SDEKernel,
                 chol_obs_covariance: Optional[torch.Tensor] = None) -> None:
        """
        :param input_data: ``(time_points [batch + [num_data]], observations [batch + [num_data, observation_dim]])``.
        :param chol_obs_covariance: ``[observation_dim, observation_dim]`` Cholesky factor of the noise covariance
            (default: identity, as in the reference).
        """
        time_points, observations = input_data
        obs_dim = observations.shape[-1]
        if chol_obs_covariance is None:
            chol_obs_covariance = torch.eye(obs_dim, dtype=observations.dtype, device=observations.device)
        if tuple(chol_obs_covariance.shape) != (obs_dim, obs_dim):
            raise ValueError("chol_obs_covariance must have shape [observation_dim, observation_dim]")
        if tuple(time_points.shape) != tuple(observations.shape[:-1]):
            raise ValueError("time_points must have shape observations.shape[:-1]")
        self._kernel = kernel
        self._time_points = time_points
        self._observations = observations
        self._chol_obs_covariance = chol_obs_covariance

    @property
    def time_points(self) -> torch.Tensor:
        return self._time_points

    @property
    def observations(self) -> torch.Tensor:
        return self._observations

    @property
    def kernel(self) -> SDEKernel:
        return self._kernel

    @property
    def _kalman(self) -> KalmanFilter:
        """gaussian_process_regression.py:112-124 (no mean function: residuals = observations)."""
        return KalmanFilter(
            state_space_model=self._kernel.state_space_model(self._time_points),
            emission_model=self._kernel.generate_emission_model(self._time_points),
            observations=self._observations,
            chol_obs_covariance=self._chol_obs_covariance,
        )

    def log_likelihood(self) -> torch.Tensor:
        """``log p(y | ϑ)`` summed over the batch (gaussian_process_regression.py:150-160)."""
        return self._kalman.log_likelihood()

    def loss(self) -> torch.Tensor:
        return -self.log_likelihood()

    def posterior_state_space_model(self) -> StateSpaceModel:
        """The smoothed chain on the training time points (what the reference's ``posterior`` is built from, :138-144)."""
        return self._kalman.posterior_state_space_model()
