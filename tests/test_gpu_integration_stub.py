"""The ctypes binding shown in INTEGRATION.md ("Replacing the process pool"), executed
as written against libssrs_hip.so -- no ssrs_amd host code in the call path -- and
checked against the oracle.  If this file has to change, INTEGRATION.md has to."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_oracle(gpu):
    import torch
    from oracle import c_oracle, ssrs_oracle as orc
    from ssrs_amd.synthetic import synthetic_dem

    # ---- inputs a reference Simulator would hold at simulator.py:346
    rows, cols = 160, 200
    z = synthetic_dem((rows, cols), 100., seed=4)
    oro = orc.compute_orographic_updraft(10., 270., orc.compute_slope_degrees(z, 100.),
                                         orc.compute_aspect_degrees(z, 100.)).astype(np.float32)
    updraft = orc.get_above_threshold_speed(oro, 0.75)
    potential = (1000. * (1 - np.arange(rows)[:, None] / (rows - 1.)) +
                 np.random.default_rng(2).normal(0, 1.0, (rows, cols))).astype(np.float32)
    rng = np.random.default_rng(3)
    starting_rows, starting_cols = rng.integers(1, 6, 500), rng.integers(0, cols, 500)
    track_direction, track_dirn_restrict, track_stochastic_nu, sim_seed, real_id = 0., 1, 1., 30, 0

    # ---- INTEGRATION.md section B preamble
    lib = C.CDLL(os.path.join(ROOT, 'ssrs_amd', 'libssrs_hip.so'))
    lib.ssrs_last_error.restype = C.c_char_p

    def dev(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a)).to('cuda', dt)

    def p(t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    def ok(rc):
        if rc:
            raise (ValueError if rc in (-1, -3) else RuntimeError)(lib.ssrs_last_error().decode())

    def STREAM():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- "Replacing the process pool"
    class P(C.Structure):                                   # SsrsTrackParams
        _fields_ = [('rows', C.c_int32), ('cols', C.c_int32), ('burnin', C.c_int32),
                    ('memory_parameter', C.c_int32), ('max_moves', C.c_int64),
                    ('scaling_parameter', C.c_double), ('prior', C.c_double * 9),
                    ('steps_per_launch', C.c_int32), ('flags', C.c_int32)]
    prm = P()
    ok(lib.ssrs_track_params_init(C.byref(prm), rows, cols, track_dirn_restrict, C.c_double(track_stochastic_nu)))
    prm.prior[:] = [float(v) for v in orc.get_directional_probs(track_direction * np.pi / 180.)]
    upd, pot = dev(updraft, torch.float64), dev(potential, torch.float32)
    lib.ssrs_transition_thr_bytes.restype = C.c_size_t
    table = torch.empty(lib.ssrs_transition_thr_bytes(rows, cols) // 4, dtype=torch.int32, device='cuda')
    ok(lib.ssrs_transition_thr_build(p(upd), p(pot), prm.prior, p(table), rows, cols, STREAM()))
    prm.flags |= 128                                        # SSRS_TRACKS_THR_TABLE
    starts = dev(np.stack([starting_rows, starting_cols], 1), torch.int32)
    n = len(starting_rows)
    lib.ssrs_tracks_workspace_bytes.restype = C.c_size_t
    nb = lib.ssrs_tracks_workspace_bytes(C.c_int64(n))
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    hist = torch.zeros((rows, cols), dtype=torch.int32, device='cuda')
    lengths = torch.empty(n, dtype=torch.int32, device='cuda')
    ends = torch.empty((n, 2), dtype=torch.int16, device='cuda')
    pool = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')     # recorded visits: 4 B per step and slot
    lib.ssrs_traj_recorder_create.restype = C.c_void_p
    rec = C.c_void_p(lib.ssrs_traj_recorder_create(p(pool), C.c_size_t(pool.numel())))
    ok(lib.ssrs_tracks_simulate_rec(C.byref(prm), p(upd), p(pot), p(table), p(starts), C.c_int64(n),
                                    C.c_uint64(sim_seed + real_id), C.c_uint64(0),   # seed, first global track id
                                    p(hist), p(ends), p(lengths), rec,
                                    p(ws), C.c_size_t(nb), None, STREAM()))
    assert lib.ssrs_traj_recorder_complete(rec)             # else: a larger pool, or the two-call form
    offsets = torch.zeros(n + 1, dtype=torch.int64, device='cuda')
    torch.cumsum(lengths, 0, out=offsets[1:])
    traj = torch.empty((int(offsets[-1]), 2), dtype=torch.int16, device='cuda')
    cursor = torch.empty(n, dtype=torch.int32, device='cuda')
    ok(lib.ssrs_tracks_gather(rec, p(starts), C.c_int64(n), p(offsets), p(traj), p(cursor), C.c_size_t(4 * n), STREAM()))
    torch.cuda.synchronize()
    lib.ssrs_traj_recorder_destroy(rec)
    off = offsets.cpu().numpy()
    tracks = [traj.cpu().numpy()[off[i]:off[i + 1]] for i in range(n)]      # the list simulator.py:382-385 pickles

    # ---- the oracle on the same inputs and streams
    ref = c_oracle.simulate_tracks(track_direction, np.stack([starting_rows, starting_cols], 1), (rows, cols),
                                   track_dirn_restrict, track_stochastic_nu, updraft, potential,
                                   seed=sim_seed + real_id, want_traj=True)
    assert np.array_equal(lengths.cpu().numpy(), ref['lengths'])
    assert np.array_equal(ends.cpu().numpy(), ref['ends'])
    assert np.array_equal(hist.cpu().numpy().view(np.uint32), ref['hist'])
    for a, b in zip(tracks, ref['tracks']):
        assert np.array_equal(a, b)

    # bad arguments come back as codes + message, not as crashes
    prm.memory_parameter = 3
    rc = lib.ssrs_tracks_simulate(C.byref(prm), p(upd), p(pot), p(table), p(starts), C.c_int64(n),
                                  C.c_uint64(1), C.c_uint64(0), p(hist), p(ends), p(lengths), None, None,
                                  p(ws), C.c_size_t(nb), None, STREAM())
    assert rc != 0 and b'THR_TABLE' in lib.ssrs_last_error()


def test_hist_reduce_through_the_c_abi_with_a_callers_rccl_communicator(gpu):
    """ssrs_hist_reduce: a C-ABI consumer brings its own ncclComm_t.  One GPU here,
    so the communicator has one rank (RCCL refuses two ranks on one device): the call must go
    through RCCL and leave the counts intact, for ncclReduce and ncclAllReduce; the multi-rank sum
    itself is RCCL's (the driver's 8-GPU run exercises it through torch.distributed)."""
    import ctypes as C
    import numpy as np
    import torch
    from ssrs_amd import _native as nat
    rccl = C.CDLL('librccl.so.1', mode=C.RTLD_GLOBAL)      # the caller's RCCL, visible to the library

    class UniqueId(C.Structure):
        _fields_ = [('internal', C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        rng = np.random.default_rng(3)
        h = rng.integers(0, 2 ** 32, size=(300, 400), dtype=np.uint64).astype(np.uint32)
        t = torch.from_numpy(h.view(np.int32).copy()).cuda()
        stream = torch.cuda.current_stream()
        for root in (0, -1):
            nat.check(nat.lib().ssrs_hist_reduce(nat.ptr(t), C.c_size_t(t.numel()), root, comm,
                                                 C.c_void_p(stream.cuda_stream)))
            stream.synchronize()
            assert np.array_equal(t.cpu().numpy().view(np.uint32), h)
        with pytest.raises(ValueError):
            nat.check(nat.lib().ssrs_hist_reduce(nat.ptr(t), C.c_size_t(t.numel()), 0, None, None))
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
