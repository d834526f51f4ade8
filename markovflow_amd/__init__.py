"""
markovflow_amd - MI355X-native Kalman filter / block-tridiagonal precision operator.

Drop-in for the hot path of secondmind-labs/markovflow (``kalman_filter.py``, ``block_tri_diag.py``,
``state_space_model.py``, ``gauss_markov.py``, ``emission_model.py``): same class and method names,
torch HIP tensors in place of TensorFlow tensors, hand-written gfx950 kernels underneath.
"""
from .block_tri_diag import BlockTriDiagonal, LowerTriangularBlockTriDiagonal, SymmetricBlockTriDiagonal
from .emission_model import EmissionModel
from .gauss_markov import GaussMarkovDistribution, check_compatible
from .kalman_filter import (
    BaseKalmanFilter,
    GaussianSites,
    KalmanFilter,
    KalmanFilterWithSites,
    KalmanFilterWithSparseSites,
    UnivariateGaussianSitesNat,
)
from .state_space_model import StateSpaceModel, state_space_model_from_covariances
from . import conditionals, distributed, kernels, models, ssm_gaussian_transformations
from .kernels import IndependentMultiOutput, Matern12, Matern32, Matern52, SDEKernel, StationaryKernel, Sum
from .models import GaussianProcessRegression
from .posterior import AnalyticPosteriorProcess, ConditionalProcess
from ._lib import MarkovflowAmdError, check_errors, errors_as_nan, set_synchronous_checks

__all__ = [
    "BlockTriDiagonal", "LowerTriangularBlockTriDiagonal", "SymmetricBlockTriDiagonal", "EmissionModel",
    "GaussMarkovDistribution", "check_compatible", "BaseKalmanFilter", "GaussianSites", "KalmanFilter",
    "KalmanFilterWithSites", "KalmanFilterWithSparseSites", "UnivariateGaussianSitesNat", "StateSpaceModel",
    "state_space_model_from_covariances", "conditionals", "distributed", "kernels", "models", "ssm_gaussian_transformations", "SDEKernel", "StationaryKernel", "Matern12", "Matern32",
    "Matern52", "Sum", "IndependentMultiOutput", "GaussianProcessRegression", "AnalyticPosteriorProcess", "ConditionalProcess",
    "MarkovflowAmdError", "check_errors", "errors_as_nan", "set_synchronous_checks",
]
