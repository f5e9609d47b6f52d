"""BASELINE configs[2] / configs[4] through the product API on ONE GPU: two fresh processes
(gloo group, both on cuda:0) run Simulator.simulate_tracks + compute_presence_map; the files
rank 0 leaves behind must equal the single-process run's bit for bit -- tracks sharded over the
ranks in uniform / snapshot mode (one case), cases sharded in seasonal mode
(/root/reference/ssrs/simulator.py:347-369 maps tracks over a pool; :200-215 loops over cases)."""
import glob
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(world, out_dir, mode, expect_rc=0, env_extra=None):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(env_extra or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'mp_simulator_worker.py'), str(r), str(world),
                               port, out_dir, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    # a rank that dies (assertion, HIP error) leaves its peer in a gloo collective: nobody may be left
    # behind holding cuda:0, whatever happens here
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0].decode('utf-8', 'replace'))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == expect_rc, f'rank {r} exited with {p.returncode}:\n{o[-3000:]}'
    if expect_rc:
        return outs
    return os.path.join(out_dir, f'{mode}_w{world}', 'data', mode if mode in ('snapshot', 'seasonal') else 'uniform')


@pytest.mark.parametrize('mode', ['uniform', 'snapshot', 'seasonal', 'subbatch'])
def test_two_ranks_equal_one_rank(gpu, tmp_path, mode):
    one = _run(1, str(tmp_path), mode)
    two = _run(2, str(tmp_path), mode)
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(one, '*')))
    assert names == sorted(os.path.basename(f) for f in glob.glob(os.path.join(two, '*'))), \
        'the two-rank run left a different set of files (stray .part files?)'
    assert any(n.endswith('_tracks.pkl') for n in names) and 'summary_presence.npy' in names
    for n in names:
        a, b = os.path.join(one, n), os.path.join(two, n)
        if n.endswith('.npy'):
            assert np.array_equal(np.load(a), np.load(b)), n
        elif n.endswith('.pkl'):
            with open(a, 'rb') as fa, open(b, 'rb') as fb:
                ta, tb = pickle.load(fa), pickle.load(fb)
            assert len(ta) == len(tb)
            for x, y in zip(ta, tb):
                assert np.array_equal(x, y), n


def test_unseeded_shards_step_under_one_seed(gpu, tmp_path):
    """sim_seed < 0: each process would draw its own start cells and stream key; the shards of one
    case are one batch, so rank 0's draws are broadcast (the merged tracks.pkl is then the run of ONE
    key: every track id occurs once, in order)."""
    import json
    out = _run(2, str(tmp_path), 'unseeded')
    seeds = []
    for r in range(2):
        with open(os.path.join(str(tmp_path), f'seeds_w2_r{r}.json')) as f:
            seeds.append(json.load(f))
    assert seeds[0] == seeds[1] and len(seeds[0]) == 1
    with open(glob.glob(os.path.join(out, '*_tracks.pkl'))[0], 'rb') as f:
        tracks = pickle.load(f)
    assert len(tracks) == 301
    # one process, the same key and the same start cells: identical trajectories
    import torch
    from ssrs_amd import movmodel
    starts = np.stack([t[0] for t in tracks]).astype(np.int32)
    dem_run = os.path.join(out, 's10d270_orograph.npy')
    from ssrs_amd import layers
    upd = layers.get_above_threshold_speed(torch.from_numpy(np.load(dem_run)).cuda(), 0.75)
    pot = np.load(glob.glob(os.path.join(out, '*_potential.npy'))[0])
    one = movmodel.simulate_tracks(0., starts, upd.shape, 1, 1., upd, pot, seed=seeds[0][0][1], want_tracks=True)
    for a, b in zip(one.tracks(), tracks):
        assert np.array_equal(a, b)


def test_file_size_guard_is_agreed_on_by_all_ranks(gpu, tmp_path):
    """max_tracks_file_gb is the limit of the MERGED <id>_tracks.pkl: with a limit between one rank's share
    and the whole every rank refuses together (a rank raising alone left the others in
    _write_tracks' barrier until the collective timed out), and one process refuses the same run."""
    import re
    one = _run(1, str(tmp_path), 'uniform')
    with open(glob.glob(os.path.join(one, '*_tracks.pkl'))[0], 'rb') as f:
        tracks = pickle.load(f)
    total = sum(len(t) for t in tracks) * 4
    half = sum(len(t) for t in tracks[:151]) * 4            # rank 0's share of 301 tracks
    assert half < 0.75 * total
    limit_gb = 0.875 * total / 2 ** 30                      # above either share, below the merged file
    assert half < limit_gb * 2 ** 30 < total
    for world in (1, 2):
        outs = _run(world, str(tmp_path), 'fileguard', expect_rc=7, env_extra={'SSRS_TEST_FILE_GB': repr(limit_gb)})
        for o in outs:
            m = re.search(r'REFUSED: .*these 301 tracks are', o)
            assert m, o[-2000:]
