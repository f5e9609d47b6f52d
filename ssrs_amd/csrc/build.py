"""Builds ssrs_amd/libssrs_hip.so for gfx950 with hipcc (cross-compiles without
a GPU).  -ffp-contract=off is REQUIRED: the stepper's move decision reproduces
the reference's f64 rounding sequence, which a fused multiply-add would break."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ['capi.hip', 'raster.hip', 'tracks.hip', 'presence.hip', 'potential.hip', 'thermals.hip', 'amg.hip']
LIB = os.path.join(PKG, 'libssrs_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fno-fast-math', '-fgpu-rdc=0' if False else '-Wall', '-Wno-unused-function']


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    deps = srcs + [os.path.join(HERE, 'amg.h'), os.path.join(HERE, 'common.h'),
                   os.path.join(os.path.dirname(PKG), 'include', 'ssrs_hip.h')]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    objs = []
    procs = []
    for s in srcs:
        o = os.path.splitext(s)[0] + '.o'
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= _newest(
                [s, deps[-1], deps[-2], deps[-3]]):     # every header: amg.h, common.h, ssrs_hip.h
            continue
        cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd))
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {s}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_probe(defines, tag):
    """Timing-probe variant of the library (tools/attic/probe_chain.py): tracks.hip recompiled with
    -D<defines>, linked with the product's other objects into libssrs_probe_<tag>.so."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    build()
    out = os.path.join(PKG, f'libssrs_probe_{tag}.so')
    obj = os.path.join(HERE, f'tracks_probe_{tag}.o')
    subprocess.check_call([hipcc] + FLAGS + [f'-D{d}' for d in defines] +
                          ['-c', os.path.join(HERE, 'tracks.hip'), '-o', obj])
    objs = [os.path.join(HERE, os.path.splitext(s)[0] + '.o') for s in SOURCES if s != 'tracks.hip'] + [obj]
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return out


def build_variant(defines, tag, sources=('amg.hip', 'potential.hip'), travel=False):
    """A/B variant of the product library: `sources` recompiled with -D<defines>, linked with the product's other
    objects into libssrs_probe_<tag>.so (SSRS_HIP_LIB selects it)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    build()
    out = os.path.join(PKG, f'libssrs_{"ab" if travel else "probe"}_{tag}.so')      # (probe libraries do not travel to the GPU box)
    objs = []
    for s in SOURCES:
        if s in sources:
            o = os.path.join(HERE, os.path.splitext(s)[0] + f'_probe_{tag}.o')
            subprocess.check_call([hipcc] + FLAGS + [f'-D{d}' for d in defines] + ['-c', os.path.join(HERE, s), '-o', o])
        else:
            o = os.path.join(HERE, os.path.splitext(s)[0] + '.o')
        objs.append(o)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + ['-ldl'])
    return out


if __name__ == '__main__':
    if '--win72' in sys.argv:         # 72-row block windows: two k_step_roam blocks per CU (A/B, results identical)
        print(build_variant(['SSRS_WIN_ROWS=72'], 'win72', sources=('tracks.hip',), travel=True))
    elif '--probe-nostray' in sys.argv:   # k_step_roam without the strays' global atomics (timing only: wrong histogram)
        print(build_variant(['SSRS_PROBE_NO_STRAY_ATOMICS'], 'probe_nostray', sources=('tracks.hip',), travel=True))
    elif '--amg-f32' in sys.argv:       # the f32 V-cycle measured and rejected in round 4 (stalls at 1e-8: amg.h)
        print(build_variant(['SSRS_AMG_CYCLE_F32'], 'amg_f32'))
    elif '--probe-k3' in sys.argv:      # binning-kernel timing probes (wrong histograms: bench timing only, no checks)
        for tag, defs in (('k3_noflush', ['SSRS_PROBE_K3_NOFLUSH']), ('k3_noread', ['SSRS_PROBE_K3_NOREAD'])):
            print(build_probe(defs, tag))
    elif '--probe-k2a' in sys.argv:     # table-builder timing probes: tools/dev/time_k2a.py ONLY (the tables are garbage)
        for tag, defs in (('k2a_nostore', ['SSRS_PROBE_K2A_NOSTORE']), ('k2a_noload', ['SSRS_PROBE_K2A_NOLOAD']), ('k2a_pad', ['SSRS_PROBE_K2A_PAD=4352']), ('k2a_pad2', ['SSRS_PROBE_K2A_PAD=1048576+4352'])):
            print(build_probe(defs, tag))
    elif '--probe' in sys.argv:
        for tag, defs in (('nophilox', ['SSRS_PROBE_NO_PHILOX']), ('nogather', ['SSRS_PROBE_NO_GATHER']),
                          ('neither', ['SSRS_PROBE_NO_PHILOX', 'SSRS_PROBE_NO_GATHER'])):
            print(build_probe(defs, tag))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
