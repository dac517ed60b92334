"""bayesnf_amd: MI355X-native ensemble training for Bayesian Neural Fields.

Drop-in for the estimator surface of google/bayesnf
(/root/reference/src/bayesnf/__init__.py:20-23); the engine underneath is a
hand-written HIP library for gfx950 (bayesnf_amd/csrc, C ABI in include/bnf.h).
"""

from .spatiotemporal import BayesianNeuralFieldMAP
from .spatiotemporal import BayesianNeuralFieldMLE
from .spatiotemporal import BayesianNeuralFieldVI

__version__ = '0.1.0'
__all__ = ['BayesianNeuralFieldMAP', 'BayesianNeuralFieldMLE', 'BayesianNeuralFieldVI']
