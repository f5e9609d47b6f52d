"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/ssrs_hip.h declares; argument validation runs before any GPU work."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'ssrs_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ssrs_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    from ssrs_amd import _native
    lib = _native.lib()
    names = declared_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f'{name} declared in ssrs_hip.h but not exported'
    assert sorted(_native.EXPORTS) == names, 'ssrs_amd._native.EXPORTS out of sync with the header'


def test_version_and_error_text():
    from ssrs_amd import _native
    lib = _native.lib()
    assert lib.ssrs_version() == 108
    assert isinstance(lib.ssrs_last_error(), bytes)


def test_threshold_table_size_includes_its_guard_bands():
    """Eight planes at a power-of-two stride between two guard bands of (cols + 2) entries, rounded up to
    256 bytes (the stepper's speculative gathers of boundary cells land there); a table too large for
    32-bit offsets is refused by the builder."""
    from ssrs_amd import _native
    lib = _native.lib()
    lib.ssrs_transition_thr_bytes.restype = C.c_size_t
    for rows, cols in ((5000, 6000), (96, 128), (5, 5)):
        stride = 256
        while stride < rows * cols * 4:
            stride *= 2
        guard = -(-(cols + 2) * 4 // 256) * 256
        assert lib.ssrs_transition_thr_bytes(rows, cols) == 8 * stride + 2 * guard
    assert lib.ssrs_transition_thr_bytes(0, 10) == 0
    dummy = (C.c_double * 9)()
    buf = (C.c_char * 64)()
    rc = lib.ssrs_transition_thr_build(buf, None, dummy, buf, 9000, 9000, None)      # 8.1e7 cells > 2^26
    assert rc == _native.SSRS_ERR_INVALID and b'2^26' in lib.ssrs_last_error()


def test_argument_validation_needs_no_gpu():
    from ssrs_amd import _native
    lib = _native.lib()
    rc = lib.ssrs_slope_aspect(None, 1, C.c_double(10.), None, None, 1, 10, 10, None)
    assert rc == _native.SSRS_ERR_INVALID and b'dem is NULL' in lib.ssrs_last_error()
    rc = lib.ssrs_threshold_updraft(None, C.c_double(0.75), None, C.c_size_t(4), None)
    assert rc == _native.SSRS_ERR_INVALID
    p = _native.SsrsTrackParams()
    assert lib.ssrs_track_params_init(C.byref(p), 500, 600, 1, C.c_double(1.0)) == 0
    assert (p.rows, p.cols, p.burnin, p.max_moves) == (500, 600, 50, 75000)
    assert lib.ssrs_track_params_init(C.byref(p), 5000, 6000, 1, C.c_double(1.0)) == 0
    assert (p.burnin, p.max_moves) == (500, 7500000)
    assert lib.ssrs_track_params_init(C.byref(p), 31, 33, 3, C.c_double(1.0)) == 0
    assert (p.burnin, p.max_moves, p.memory_parameter) == (3, 256, 3)    # ceil(255.75)
    assert lib.ssrs_track_params_init(C.byref(p), 3, 600, 1, C.c_double(1.0)) == _native.SSRS_ERR_INVALID
    p.memory_parameter = 9
    rc = lib.ssrs_tracks_simulate(C.byref(p), None, None, None, None, C.c_int64(1), C.c_uint64(0),
                                  C.c_uint64(0), None, None, None, None, None, None,
                                  C.c_size_t(0), None, None)
    assert rc == _native.SSRS_ERR_INVALID
    with pytest.raises(ValueError):
        _native.check(rc)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import numpy as np
    from ssrs_amd import layers
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        layers.compute_slope_degrees(np.zeros((8, 8)), 10.)


def test_ctypes_structs_match_the_header(tmp_path):
    """The header's structs, compiled as plain C (gcc), have the size and field offsets of
    their ctypes mirrors in ssrs_amd/_native.py."""
    import ctypes, shutil, subprocess
    from ssrs_amd import _native as nat
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    structs = {'SsrsTrackParams': nat.SsrsTrackParams, 'SsrsTrackStats': nat.SsrsTrackStats,
               'SsrsSolveStats': nat.SsrsSolveStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ssrs_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({name}, {field}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / 'abi.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'abi'
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', inc, str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line in out:
        name, size, *offsets = line.split()
        cls = structs[name]
        assert int(size) == ctypes.sizeof(cls), name
        assert [int(o) for o in offsets] == [getattr(cls, f).offset for f, _ in cls._fields_], name


def test_graft_entry_build_checks_the_headers_version():
    """__graft_entry__.build() compares the library with the version include/ssrs_hip.h declares (it used to carry a literal
    that two ABI bumps of round 4 left behind: the driver's build check would have failed)."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, '__graft_entry__.py')).read()
    assert 'SSRS_VERSION' in src and not re.search(r'ssrs_version\(\)\s*==\s*\d', src)
    from ssrs_amd import _native
    declared = int(re.search(r'#define\s+SSRS_VERSION\s+(\d+)', open(os.path.join(root, 'include', 'ssrs_hip.h')).read()).group(1))
    assert _native.lib().ssrs_version() == declared
