"""
Emission model ``f_k = H_k x_k`` - mirror of ``markovflow/emission_model.py:25-153`` (reference).
The projections are tiny batched products; inside ``KalmanFilter.log_likelihood`` they are fused
into the HIP kernel and never materialised.
"""
from typing import Tuple

import torch


class EmissionModel:
    """Linear projection of states to outputs (emission_model.py:25-153)."""

    def __init__(self, emission_matrix: torch.Tensor) -> None:
        """:param emission_matrix: ``batch_dim + [num_data, output_dim, state_dim]``."""
        if emission_matrix.dim() < 3:
            raise ValueError(
                f"Emission Matrix must be at least 3D but has shape {tuple(emission_matrix.shape)}"
            )  # emission_model.py:46-49
        self._H = emission_matrix

    @property
    def batch_shape(self) -> torch.Size:
        return self._H.shape[:-3]

    @property
    def num_data(self) -> int:
        return self._H.shape[-3]

    @property
    def output_dim(self) -> int:
        return self._H.shape[-2]

    @property
    def state_dim(self) -> int:
        return self._H.shape[-1]

    @property
    def emission_matrix(self) -> torch.Tensor:
        return self._H

    def project_state_marginals_to_f(
        self, means: torch.Tensor, covariances: torch.Tensor, full_output_cov: bool = False
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.project_state_to_f(means), self.project_state_covariance_to_f(covariances, full_output_cov)

    def project_state_to_f(self, state: torch.Tensor) -> torch.Tensor:
        """``H x`` (emission_model.py:115-128)."""
        if state.shape[-1] != self.state_dim or state.shape[-2] != self.num_data:
            raise ValueError(f"state has shape {tuple(state.shape)}, expected [..., {self.num_data}, {self.state_dim}]")
        return torch.matmul(self._H, state[..., None])[..., 0]

    def project_state_covariance_to_f(self, covariance: torch.Tensor, full_output_cov: bool = False) -> torch.Tensor:
        """``H S Hᵀ`` or its diagonal (emission_model.py:130-153)."""
        if tuple(covariance.shape[-3:]) != (self.num_data, self.state_dim, self.state_dim):
            raise ValueError(f"covariance has shape {tuple(covariance.shape)}")
        hs = torch.matmul(self._H, covariance)
        if full_output_cov:
            return torch.matmul(hs, self._H.transpose(-1, -2))
        return torch.sum(self._H * hs, dim=-1)
