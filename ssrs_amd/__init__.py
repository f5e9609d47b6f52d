"""ssrs_amd -- MI355X-native implementation of the SSRS data-parallel hot path
(updraft raster + stochastic track stepper) behind the reference's
`Config` / `Simulator` API.  Host code is Python; all numeric work runs in
hand-written HIP kernels (libssrs_hip.so, include/ssrs_hip.h)."""
from .config import Config
from .simulator import Simulator
from .layers import (compute_orographic_updraft, compute_slope_degrees,
                     compute_aspect_degrees, get_above_threshold_speed)

__all__ = ['Config', 'Simulator', 'compute_orographic_updraft', 'compute_slope_degrees',
           'compute_aspect_degrees', 'get_above_threshold_speed']
