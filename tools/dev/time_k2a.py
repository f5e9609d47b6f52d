"""Time the threshold-table builder alone (SSRS_HIP_LIB selects the library; probe libraries build
garbage tables and must never be stepped on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssrs_amd import movmodel
rows, cols = 5000, 6000
g = torch.Generator(device='cuda').manual_seed(1)
upd = torch.rand((rows, cols), device='cuda', dtype=torch.float64, generator=g) * 2
pot = (1000. * (1 - torch.arange(rows, device='cuda', dtype=torch.float64)[:, None] / (rows - 1.)) + torch.rand((rows, cols), device='cuda', dtype=torch.float64, generator=g)).float()
for _ in range(3):
    t = movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    t = movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(os.environ.get('SSRS_HIP_LIB', 'product'), f'{ms * 1e3:.1f} us  {rows * cols * 44 / ms / 1e9:.2f} TB/s (44 B/cell)')
