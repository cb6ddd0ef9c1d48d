"""CPU-side checks of the drop-in boundary: libbdof.so loads and exports every symbol that
include/bdof.h declares; the product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

import __graft_entry__ as entry
from beyond_dof_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    entry.build()
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'bdof.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bdof_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert _header_symbols() == _lib.EXPORTED_SYMBOLS


def test_library_exports_every_declared_symbol(built_lib):
    for name in _header_symbols():
        assert hasattr(built_lib, name), name


def test_no_cpu_fallback(built_lib):
    """Without a HIP device the engine must fail loudly (bdof_device_count() == 0 here)."""
    if built_lib.bdof_device_count() > 0:
        pytest.skip('a GPU is present')
    from beyond_dof_amd.engine import MultisliceEngine
    with pytest.raises(_lib.BdofError):
        MultisliceEngine(64, 64, 4, 1)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'beyond_dof_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no CPU fallback', ''), os.path.join(dirpath, f)


def test_c_example_compiles_and_links(tmp_path):
    """examples/cabi_forward.c is plain C against include/bdof.h: it must build with gcc alone and link to libbdof.so."""
    import subprocess
    import __graft_entry__ as entry
    entry.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'cabi_forward')
    subprocess.check_call(['gcc', '-O2', '-Wall', '-Werror', '-std=c99', '-I', os.path.join(root, 'include'),
                           os.path.join(root, 'examples', 'cabi_forward.c'), '-o', exe,
                           '-L', os.path.join(root, 'beyond_dof_amd'), '-lbdof', '-Wl,-rpath,' + os.path.join(root, 'beyond_dof_amd')])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b'usage' in r.stderr


def test_bench_self_launch_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` on a host without a GPU: the parent starts two ranks, they rendezvous, find no device and raise
    (no CPU fallback); the parent reports the failure promptly, prints no JSON line and leaves no rank behind."""
    import subprocess
    import sys
    lib = _lib.load()
    if lib.bdof_device_count() > 0:
        pytest.skip('a GPU is present')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '64', '--angles-per-gpu', '2', '--n-theta', '4',
                        '--steps', '1', '--warmup', '0', '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0
    assert b'no CPU fallback' in r.stderr
    assert r.stdout.strip() == b''


def test_no_spill_under_restricted_exec():
    """The gfx950 ISA of every kernel holds no register spill that is stored just before `s_or_b64 exec, exec, ...` restores the
    mask of a divergent region (tools/check_spills.py): hipcc 7.2 produced two in k_resident<128> in round 3, and the waves that
    had no lane in the region reloaded stale scratch — a loss 5 % off, depending on what had run on the CU before.  Compiles the
    device code once (about a minute, no GPU needed)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_spills
    isa = check_spills.device_isa()
    hits = check_spills.scan(isa)
    assert not hits, hits[:5]
    # the same listing: k_conv2's counted wait (csrc/bdof_conv2.h) — `s_waitcnt vmcnt(N)` by asm at the end of a tile means
    # "everything but this tile's N stores has retired" only if the N youngest vector-memory operations there are stores
    bad, n = check_spills.scan_counted_waits(isa)
    assert n >= 12 and not bad, (n, bad[:5])          # 3 tap counts x forward / backward x scalar / field carrier


@pytest.mark.parametrize('n', [64, 512, 1024])
def test_dithered_twiddle_tables(built_lib, n):
    """The transform constants of the per-slice kernels (DESIGN §5 "Dithered transform constants"; host code, no device): in each
    of the D copies every component is one of the two float32 neighbours of the float64 value, hi + lo of every copy is that value,
    and the mean over ANY run of L consecutive copies (the slices a wave passes through) is the value to ulp / L — where one
    round-to-nearest table is off by up to ulp / 2 in every slice."""
    import ctypes
    import numpy as np
    D = 64
    buf = np.zeros((D, 2, n, 2), dtype=np.float32)
    assert built_lib.bdof_twiddle_tables(n, D, buf.ctypes.data_as(ctypes.c_void_p)) == 0
    j = np.arange(n)
    exact = np.stack([np.cos(-2 * np.pi * j / n), np.sin(-2 * np.pi * j / n)], axis=-1)          # float64
    hi = buf[:, 0].astype(np.float64)
    lo = buf[:, 1].astype(np.float64)
    below = np.nextafter(exact.astype(np.float32), np.float32(-4)).astype(np.float64)
    below = np.where(exact.astype(np.float32).astype(np.float64) <= exact, exact.astype(np.float32).astype(np.float64), below)
    above = np.nextafter(below.astype(np.float32), np.float32(4)).astype(np.float64)
    ulp = above - below
    assert np.all((hi == below) | (hi == above) | (exact == below))
    assert np.abs(hi + lo - exact).max() <= 2e-15                                                  # the pair is exact
    for L in (8, 16, 64):
        for start in (0, 5, 40):
            idx = (start + np.arange(L)) % D
            err = np.abs(hi[idx].mean(axis=0) - exact) / ulp
            assert err.max() <= 1.0 / L + 1e-12, (L, start, err.max())
    # one nearest-rounded table for comparison: its error does not shrink with the number of slices
    one = np.zeros((1, 2, n, 2), dtype=np.float32)
    assert built_lib.bdof_twiddle_tables(n, 0, one.ctypes.data_as(ctypes.c_void_p)) == 0
    assert (np.abs(one[0, 0].astype(np.float64) - exact) / ulp).max() > 0.4


def test_counted_wait_scanner_flags_a_wait_not_backed_by_stores():
    """tools/check_spills.scan_counted_waits on a made-up listing: two stores before vmcnt(2) pass, a load among them or a block
    boundary before the second store is reported."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_spills
    head = '_Z7k_conv2ILb0ELi8ELb0ELi64EEv8ConvArgs:  ; @k\n.LBB0_1:\n'
    wait = '\t;;#ASMSTART\n\ts_waitcnt vmcnt(2)\n\t;;#ASMEND\n'
    good = head + '\tglobal_store_dwordx2 v[0:1], v[2:3], off\n\tv_add_u32_e32 v0, v1, v2\n\tglobal_store_dwordx2 v[4:5], v[2:3], off\n' + wait
    assert check_spills.scan_counted_waits(good) == ([], 1)
    load = head + '\tglobal_store_dwordx2 v[0:1], v[2:3], off\n\tglobal_load_dword v9, v[2:3], off\n\tglobal_store_dwordx2 v[4:5], v[2:3], off\n' + wait
    bad, n = check_spills.scan_counted_waits(load)
    assert n == 1 and len(bad) == 1 and 'global_load_dword' in bad[0][2]
    split = head + '\tglobal_store_dwordx2 v[0:1], v[2:3], off\n.LBB0_2:\n\tglobal_store_dwordx2 v[4:5], v[2:3], off\n' + wait
    bad, n = check_spills.scan_counted_waits(split)
    assert len(bad) == 1 and 'only 1' in bad[0][2]
    none = head + '\tglobal_store_dwordx2 v[0:1], v[2:3], off\n\ts_endpgm\n'
    assert check_spills.scan_counted_waits(none)[0] == [('_Z7k_conv2ILb0ELi8ELb0ELi64EEv8ConvArgs', 0, 'no counted wait found')]
