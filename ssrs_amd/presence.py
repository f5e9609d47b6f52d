"""Presence density (K3'/K4) host side, behind the reference's function names
(/root/reference/ssrs/movmodel.py:410-439) plus the normalisation ladder of
Simulator.plot_presence_map (simulator.py:520-546)."""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from ._device import device, stream_ptr, to_dev, is_tensor, like_input


def _scratch(dev):
    return torch.zeros(8, dtype=torch.uint8, device=dev)


def compute_presence_counts(tracks, gridshape):
    """movmodel.py:410-419.  `tracks`: list of int16 (n_i, 2) arrays (numpy) or
    one CUDA int16 (N, 2) tensor of concatenated points.  Returns an int32
    raster holding uint32 counts (the reference's int16 wraps above 32767)."""
    rows, cols = int(gridshape[0]), int(gridshape[1])
    if is_tensor(tracks):
        pts = to_dev(tracks, torch.int16).reshape(-1, 2)
        as_numpy = False
    else:
        as_numpy = True
        tracks = [np.asarray(t, dtype=np.int16).reshape(-1, 2) for t in tracks]
        flat = np.concatenate(tracks) if tracks else np.zeros((0, 2), dtype=np.int16)
        pts = to_dev(flat, torch.int16)
    hist = torch.zeros((rows, cols), dtype=torch.int32, device=device())
    nat.check(nat.lib().ssrs_presence_count(
        nat.ptr(pts), C.c_int64(int(pts.shape[0])), nat.ptr(hist), rows, cols,
        nat.ptr(_scratch(hist.device)), stream_ptr()))
    return hist.cpu().numpy() if as_numpy else hist


def smooth_presence_counts(count_mat, radius):
    """Disk smoothing of a count matrix (movmodel.py:431-439) -> f32.  int64 counts (the
    widened sum of distributed.reduce_histogram) take the 64-bit kernel, everything else is
    read as uint32-in-int32."""
    cnt = to_dev(count_mat)
    wide = cnt.dtype == torch.int64
    if not wide and cnt.dtype != torch.int32:
        cnt = cnt.to(torch.int32)
    rows, cols = int(cnt.shape[0]), int(cnt.shape[1])
    krad = int(radius)
    out = torch.empty((rows, cols), dtype=torch.float32, device=cnt.device)
    nbytes = nat.lib().ssrs_presence_workspace_bytes(rows, cols, krad)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cnt.device)
    fn = nat.lib().ssrs_presence_smooth_u64 if wide else nat.lib().ssrs_presence_smooth
    nat.check(fn(
        nat.ptr(cnt), krad, nat.ptr(out), rows, cols, nat.ptr(ws), C.c_size_t(nbytes),
        stream_ptr()))
    return like_input(out, count_mat)


def compute_smooth_presence_counts(tracks, gridshape, radius):
    """movmodel.py:422-439: histogram + disk smoothing -> f32 (rows, cols)."""
    hist = compute_presence_counts(tracks, gridshape)
    return smooth_presence_counts(hist, radius)


def presence_kernel_radius(radius_m, resolution, gridsize):
    """simulator.py:520 (+ the int(round()) of :530)."""
    krad = min(max(radius_m / resolution, 2), min(gridsize) / 2)
    return int(round(krad))


def normalise_add(src, acc):
    """acc += src / max(src) on the device (simulator.py:531-532, :538-539)."""
    s = to_dev(src)
    if s.dtype not in (torch.float32, torch.float64):
        s = s.to(torch.float64)
    nat.check(nat.lib().ssrs_presence_normalise_add(
        nat.ptr(s), nat.SSRS_F64 if s.dtype == torch.float64 else nat.SSRS_F32,
        nat.ptr(acc), C.c_size_t(s.numel()), nat.ptr(_scratch(s.device)), stream_ptr()))
    return acc


def normalise_to_f32(src):
    """f32(src / max(src)) (simulator.py:544-546)."""
    s = to_dev(src, torch.float64)
    out = torch.empty(s.shape, dtype=torch.float32, device=s.device)
    nat.check(nat.lib().ssrs_presence_normalise_f32(
        nat.ptr(s), nat.ptr(out), C.c_size_t(s.numel()), nat.ptr(_scratch(s.device)),
        stream_ptr()))
    return out
