"""`genjax.inference.smc` (reference: src/genjax/inference/smc.py:15-27 exports ChangeTarget,
Importance, ImportanceK, SMCAlgorithm), plus what the north star adds on top of the reference:
`ParticleCollection.resample` and the fused bootstrap-SMC driver."""

from .._amd.inference import (ChangeTarget, Importance, ImportanceK, ParticleCollection, SMCAlgorithm,
                              stack_to_first_dim)
from .._amd.smc_fused import BootstrapSMC, DiscreteHMM, LinearGaussianSSM, SMCResult, StateSpaceModel

__all__ = ["ChangeTarget", "Importance", "ImportanceK", "SMCAlgorithm", "ParticleCollection", "BootstrapSMC",
           "LinearGaussianSSM", "DiscreteHMM", "StateSpaceModel", "SMCResult", "stack_to_first_dim"]
