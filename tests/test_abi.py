"""CPU: the C-ABI library builds, loads and exports every symbol include/margipose_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'margipose_hip.h')).read()
    return sorted(set(re.findall(r'^int\s+(mpose_\w+)\s*\(', src, flags=re.M)))


def test_header_declares_entry_points():
    names = declared_symbols()
    assert 'mpose_softmax_dsnt_fwd' in names and 'mpose_conv_fwd' in names and len(names) >= 20


def test_library_builds_and_exports_everything():
    from margipose_amd import build
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, 'symbols declared in include/margipose_hip.h but not exported: %s' % missing
    lib.mpose_abi_version.restype = ctypes.c_int
    from margipose_amd import _lib
    assert lib.mpose_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    import pytest
    import torch
    from margipose_amd import _lib, dsntnn
    with pytest.raises(_lib.MposeError):
        dsntnn.flat_softmax(torch.zeros(1, 17, 32, 32))
    with pytest.raises(_lib.MposeError):
        dsntnn.dsnt(torch.zeros(1, 17, 32, 32))


def test_no_kernel_spills_to_scratch():
    """Policy: no gfx950 kernel may use scratch memory (a spilled instantiation of the conv kernel once produced
    wrong results and, in the K loop, spills also force early vmcnt waits)."""
    from margipose_amd import build
    if build.is_stale() or not os.path.exists(build.LOG_PATH) or os.path.getmtime(build.LOG_PATH) < os.path.getmtime(build.LIB_PATH) - 600:
        build.build(force=True)          # (the build keeps hipcc's kernel-resource-usage remarks: one compile serves both checks)
    text = open(build.LOG_PATH).read()
    for src in build.sources():
        part = text.split('==== %s\n' % os.path.basename(src))[1].split('\n==== ')[0]
        sizes = [int(x) for x in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', part)]
        assert sizes, 'no resource-usage remarks for %s' % src
        assert max(sizes) == 0, '%s: a kernel spills %d bytes/lane to scratch' % (os.path.basename(src), max(sizes))


def test_graft_entry_build_runs():
    """The driver's "does it build" hook: compiles every HIP source and imports the package (no GPU needed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('__graft_entry__', os.path.join(ROOT, '__graft_entry__.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()


def test_launch_plan_lifecycle_without_a_device():
    """csrc/plan.hip's host side needs no GPU until something is launched: begin / recording / break / end / size / replay of an
    empty plan / destroy, one recording at a time, end without begin refused."""
    from margipose_amd import build
    lib = ctypes.CDLL(build.build())
    streams = (ctypes.c_void_p * 2)(None, ctypes.c_void_p(8).value)
    plan = ctypes.c_void_p()
    assert lib.mpose_plan_recording() == 0
    assert lib.mpose_plan_end(ctypes.byref(plan)) != 0                  # nothing is being recorded
    assert lib.mpose_plan_begin(streams, 2) == 0
    assert lib.mpose_plan_recording() == 1
    assert lib.mpose_plan_begin(streams, 2) != 0                        # one recording at a time
    assert lib.mpose_plan_break() == 0
    assert lib.mpose_plan_end(ctypes.byref(plan)) == 0 and plan.value
    assert lib.mpose_plan_recording() == 0
    n = [ctypes.c_int(-1) for _ in range(3)]
    assert lib.mpose_plan_size(plan, ctypes.byref(n[0]), ctypes.byref(n[1]), ctypes.byref(n[2])) == 0
    assert [v.value for v in n] == [0, 0, 1]
    nxt = ctypes.c_int(-1)
    assert lib.mpose_plan_replay(plan, streams, 3, 0, ctypes.byref(nxt)) != 0      # stream count differs from the recording's
    assert lib.mpose_plan_destroy(plan) == 0
    assert lib.mpose_plan_begin(streams, 2) == 0 and lib.mpose_plan_abort() == 0 and lib.mpose_plan_recording() == 0
