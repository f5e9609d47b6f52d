"""Decompose the stepper's time per step: the product library and three timing
probes of it (no Philox, no table gather, neither; SSRS_HIP_LIB picks the library, one process
each) on the bench workload: 16 384 tracks (one wave per CU: the latency of a lone wave's step)
and 100 000 tracks (the bench batch).  Results of the probe libraries are wrong on purpose."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from ssrs_amd import layers, movmodel
    from ssrs_amd.synthetic import synthetic_dem, ramp_potential
    shape = (5000, 6000)
    dem = torch.from_numpy(synthetic_dem(shape, 10.)).cuda()
    pot = torch.from_numpy(ramp_potential(shape)).cuda()
    _, upd = layers.updraft_from_dem(dem, 10., 10., 270., threshold=0.75)
    table = movmodel.build_transition_table(upd, pot, ring=True) if os.environ.get('PROBE_TABLE') == 'ring' else movmodel.build_transition_table(upd, pot, thr=True, move_dirn=0.)
    np.random.seed(30)
    r, c = movmodel.get_starting_indices(100000, (5, 55, 1, 2), 'random', (60., 50.), 10.)
    starts = np.stack([r, c], 1)
    out = {}
    for n in (16384, 100000):
        for hist in (False, True):
            best = None
            for _ in range(3):
                o = movmodel.simulate_tracks(0., starts[:n], shape, 1, 1., upd, pot, seed=30, table=table, profile=True,
                                             want_hist=hist)
                L = o.lengths.cpu().numpy() - 1
                rec = dict(kernel_ms=o.stats['kernel_ms'], launches=o.stats['launches'], steps=o.stats['total_steps'],
                           mean_steps=float(L.mean()), us_per_iteration=o.stats['kernel_ms'] * 1e3 / (o.stats['launches'] * 512),
                           ns_per_wave_step=o.stats['kernel_ms'] * 1e6 / max(o.stats['total_steps'] / 64, 1) )
                if best is None or rec['kernel_ms'] < best['kernel_ms']:
                    best = rec
            out[f'{n}{"_hist" if hist else ""}'] = best
    print('RESULT ' + json.dumps(out))
    sys.exit(0)
libs = [('product', None), ('no Philox', 'nophilox'), ('no gather', 'nogather'), ('neither', 'neither')]
rows = []
for name, tag in libs:
    env = dict(os.environ)
    if tag:
        env['SSRS_ALLOW_PROBE_LIB'] = '1'
        env['SSRS_HIP_LIB'] = os.path.join(ROOT, 'ssrs_amd', f'libssrs_probe_{tag}.so')
    p = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
    if not line:
        print(name, 'FAILED', p.stderr[-2000:])
        continue
    rows.append((name, json.loads(line[0][7:])))
print('| library | batch | stepper kernels ms | launches | steps/track | us per loop iteration (kernel ms / (launches x 512)) |')
print('|---|---|---:|---:|---:|---:|')
for name, r in rows:
    for key, v in r.items():
        print(f'| {name} | {key} | {v["kernel_ms"]:.3f} | {v["launches"]} | {v["mean_steps"]:.0f} | {v["us_per_iteration"]:.3f} |')
