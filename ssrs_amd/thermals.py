"""Thermal updraft realisations (a5) on the device, behind the reference's
`compute_thermals(aspect, thermal_intensity_scale)` (/root/reference/ssrs/
layers.py:188-214).  Statistical parity only: see csrc/thermals.hip."""
import ctypes as C

import torch

from . import _native as nat
from ._device import stream_ptr, to_dev, like_input


def gaussian_blur(field, sigma):
    """scipy.ndimage.gaussian_filter(field, sigma, mode='constant') in f64."""
    x = to_dev(field, torch.float64)
    rows, cols = int(x.shape[0]), int(x.shape[1])
    out = torch.empty_like(x)
    nbytes = nat.lib().ssrs_blur_workspace_bytes(rows, cols, C.c_double(sigma))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    nat.check(nat.lib().ssrs_gaussian_blur(nat.ptr(x), nat.ptr(out), C.c_double(sigma), rows, cols,
                                           nat.ptr(ws), C.c_size_t(nbytes), stream_ptr()))
    return like_input(out, field)


def thermal_seeds(aspect, thermal_intensity_scale, seed=0):
    a = to_dev(aspect, torch.float64)
    rows, cols = int(a.shape[0]), int(a.shape[1])
    out = torch.empty_like(a)
    nat.check(nat.lib().ssrs_thermal_seeds(nat.ptr(a), C.c_double(thermal_intensity_scale),
                                           C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                           nat.ptr(out), rows, cols, stream_ptr()))
    return like_input(out, aspect)


def compute_thermals(aspect, thermal_intensity_scale, seed=0):
    """Field of smoothed random thermals (f64), one realisation per `seed`."""
    a = to_dev(aspect, torch.float64)
    return like_input(gaussian_blur(thermal_seeds(a, thermal_intensity_scale, seed), 4.0), aspect)
