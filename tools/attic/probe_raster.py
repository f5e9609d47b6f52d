"""Timings of the K1 raster kernels at C2 (5000 x 6000), HIP events via torch."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from ssrs_amd import layers
from ssrs_amd.synthetic import synthetic_dem
rows, cols, res = 5000, 6000, 10.
dem = torch.from_numpy(synthetic_dem((rows, cols), res)).cuda()
slope, aspect = layers.slope_aspect(dem, res)
s32, a32 = slope.float(), aspect.float()
ncell = rows * cols
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cases = {
  'fused DEM f64 -> oro f32 + usable f64 (20 B/cell)': (lambda: layers.updraft_from_dem(dem, res, 10., 270., threshold=0.75), 20),
  'fused DEM f64 -> oro f32 (12 B/cell)': (lambda: layers.updraft_from_dem(dem, res, 10., 270.), 12),
  'fused DEM f32 -> oro f32 (8 B/cell)': (lambda d=dem.float(): layers.updraft_from_dem(d, res, 10., 270.), 8),
  'slope+aspect from DEM f64 -> 2 x f64 (24 B/cell)': (lambda: layers.slope_aspect(dem, res), 24),
  'orographic f32 slope/aspect -> f32 (12 B/cell)': (lambda: layers.orographic_updraft(10., 270., s32, a32), 12),
  'orographic f64 slope/aspect -> f32 (20 B/cell)': (lambda: layers.orographic_updraft(10., 270., slope, aspect), 20),
  'orographic f32 + threshold -> f32 + f64 (20 B/cell)': (lambda: layers.orographic_updraft(10., 270., s32, a32, threshold=0.75), 20),
  'threshold f32 -> f64 (12 B/cell)': (lambda o=layers.updraft_from_dem(dem, res, 10., 270.)[0]: layers.get_above_threshold_speed(o, 0.75), 12),
}
for name, (fn, b) in cases.items():
    ms = timeit(fn)
    print(f'{name:55s} {ms*1e3:8.1f} us  {ncell/ms/1e3:9.0f} Mcells/s  {ncell*b/ms/1e6:7.0f} GB/s')
