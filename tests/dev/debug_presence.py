import sys; sys.path.insert(0,'.')
import numpy as np, torch
from ssrs_amd import presence
from oracle import ssrs_oracle as orc
rng=np.random.default_rng(0)
cnt = rng.integers(0,20,(60,80)).astype(np.int32)
cnt[rng.random((60,80))<0.7]=0
sm = presence.smooth_presence_counts(torch.from_numpy(cnt).cuda(), 3)
ref = orc.smooth_presence_from_counts(cnt.astype(np.int64), 3)
print('smooth diff', np.abs(sm.cpu().numpy()-ref).max())
acc = torch.zeros((60,80),dtype=torch.float64,device='cuda')
presence.normalise_add(sm, acc)
print('case diff', np.abs(acc.cpu().numpy()-ref/ref.max()).max(), float(acc.max()))
summ = torch.zeros((60,80),dtype=torch.float64,device='cuda')
presence.normalise_add(acc, summ)
print('summ diff', np.abs(summ.cpu().numpy()-ref/ref.max()).max(), float(summ.max()))
out = presence.normalise_to_f32(summ).cpu().numpy()
print('out diff', np.abs(out-ref/ref.max()).max())
