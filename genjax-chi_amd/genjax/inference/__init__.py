"""`genjax.inference` (reference: src/genjax/inference/__init__.py:15-37)."""

from .._amd.inference import Algorithm, Marginal, SampleDistribution, Target, marginal
from . import requests, smc

__all__ = ["Algorithm", "Marginal", "SampleDistribution", "Target", "marginal", "requests", "smc"]
