"""
Abstract Gauss-Markov chain distribution - mirror of ``markovflow/gauss_markov.py`` (reference
lines 29-217): same members, same meaning; tensors are torch HIP tensors.
"""
import abc
from typing import Tuple

import torch

from .block_tri_diag import SymmetricBlockTriDiagonal

SampleShape = Tuple[int, ...]


class GaussMarkovDistribution(abc.ABC):
    """Abstract class for a Gauss-Markov chain (gauss_markov.py:29-201)."""

    @property
    @abc.abstractmethod
    def event_shape(self) -> Tuple[int, int]:
        """``[num_transitions + 1, state_dim]``."""

    @property
    @abc.abstractmethod
    def batch_shape(self) -> torch.Size:
        """Leading dims before :attr:`event_shape`."""

    @property
    @abc.abstractmethod
    def state_dim(self) -> int:
        """State dimension."""

    @property
    @abc.abstractmethod
    def num_transitions(self) -> int:
        """Number of transitions."""

    @abc.abstractmethod
    def _build_precision(self) -> SymmetricBlockTriDiagonal:
        """Compact block representation of the precision."""

    @property
    def precision(self) -> SymmetricBlockTriDiagonal:
        """Precision of the joint Gaussian (gauss_markov.py:72-77); recomputed on every access like the reference."""
        return self._build_precision()

    @property
    @abc.abstractmethod
    def marginal_means(self) -> torch.Tensor:
        """``batch_shape + [num_transitions + 1, state_dim]``."""

    @property
    @abc.abstractmethod
    def marginal_covariances(self) -> torch.Tensor:
        """``batch_shape + [num_transitions + 1, state_dim, state_dim]``."""

    @abc.abstractmethod
    def covariance_blocks(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Diagonal and lower off-diagonal blocks of the covariance."""

    @property
    def marginals(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Marginal means and covariances (gauss_markov.py:107-117)."""
        return self.marginal_means, self.marginal_covariances

    @abc.abstractmethod
    def sample(self, sample_shape: SampleShape) -> torch.Tensor:
        """``sample_shape + batch_shape + event_shape``."""

    @abc.abstractmethod
    def log_det_precision(self) -> torch.Tensor:
        """``batch_shape``."""

    @abc.abstractmethod
    def log_pdf(self, states) -> torch.Tensor:
        """``sample_shape + batch_shape``."""

    @abc.abstractmethod
    def create_trainable_copy(self) -> "GaussMarkovDistribution":
        """Copy whose parameters require gradients."""

    @abc.abstractmethod
    def create_non_trainable_copy(self) -> "GaussMarkovDistribution":
        """Detached copy."""

    @abc.abstractmethod
    def kl_divergence(self, dist: "GaussMarkovDistribution") -> torch.Tensor:
        """``KL(self || dist)`` with shape ``batch_shape``."""


def check_compatible(dist_1: GaussMarkovDistribution, dist_2: GaussMarkovDistribution) -> None:
    """Raise if two distributions are not compatible (gauss_markov.py:204-217)."""
    assert isinstance(dist_2, type(dist_1)), TypeError("`dist_2` has different representation than `dist_1`")
    if dist_1.state_dim != dist_2.state_dim:
        raise ValueError(f"state_dim mismatch: {dist_1.state_dim} vs {dist_2.state_dim}")
    if tuple(dist_1.batch_shape) != tuple(dist_2.batch_shape):
        raise ValueError(f"batch_shape mismatch: {tuple(dist_1.batch_shape)} vs {tuple(dist_2.batch_shape)}")
    if dist_1.num_transitions != dist_2.num_transitions:
        raise ValueError(f"num_transitions mismatch: {dist_1.num_transitions} vs {dist_2.num_transitions}")
