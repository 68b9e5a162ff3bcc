"""vegasafterglow_amd -- MI355X-native engine for VegasAfterglow's forward-model hot path.

Drop-in names for the accelerated path of the reference's Python API
(``Model.flux_density_grid / flux_density / flux`` and the batched log-likelihood).
"""
from . import _lib, extinction, units
from .model import (ISM, Flux, FluxDict, GaussianJet, Magnetar, MagnetizedTophatJet, Model, Observer, PowerLawJet, PowerLawWing,
                    Radiation, StepPowerLawJet, TophatJet, TwoComponentJet, Wind, coalescing_stats, get_context, set_coalescing)
from .fitting import logscale_screen

__all__ = ["ISM", "Wind", "TophatJet", "GaussianJet", "PowerLawJet", "TwoComponentJet", "StepPowerLawJet", "PowerLawWing", "MagnetizedTophatJet", "Magnetar", "Observer", "Radiation",
           "Model", "Flux", "FluxDict", "get_context", "set_coalescing", "coalescing_stats", "logscale_screen", "extinction", "units"]
__version__ = "0.1.0"
