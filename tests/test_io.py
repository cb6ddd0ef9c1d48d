"""HDF5 / TIFF boundary formats (SURVEY.md §8b "Files"): the built-in reader against files written by h5py 3.3.0
(fixtures G7), the built-in writer against the built-in reader, TIFF round trips.  CPU only."""
import os

import numpy as np
import pytest

from beyond_dof_amd import h5io, tiffio


@pytest.fixture()
def no_h5py(monkeypatch):
    monkeypatch.setattr(h5io, '_h5py', None)       # exercise the built-in parser even where h5py exists


@pytest.mark.parametrize('stem', ['g7_fullfield_3x8x8', 'g7_ptycho_2x3x4x6'])
def test_reader_on_h5py_files(golden_dir, no_h5py, stem):
    ref = np.load(os.path.join(golden_dir, stem + '.npy'))
    arr = h5io.read_dataset(os.path.join(golden_dir, stem + '.h5'))
    assert arr.dtype == np.complex64 and arr.shape == ref.shape
    assert np.array_equal(arr, ref)


def test_lazy_fancy_indexing(golden_dir, no_h5py):
    """prj[this_i_theta, this_ind_rank] of cnn_propagator/ptychography.py:295."""
    ref = np.load(os.path.join(golden_dir, 'g7_ptycho_2x3x4x6.npy'))
    f = h5io.File(os.path.join(golden_dir, 'g7_ptycho_2x3x4x6.h5'))
    d = f['exchange/data']
    assert d.shape == (2, 3, 4, 6) and len(d) == 2
    assert np.array_equal(d[1, [0, 2]], ref[1, [0, 2]])
    with pytest.raises(KeyError):
        f['exchange/nothing']


@pytest.mark.parametrize('dtype', [np.complex64, np.complex128, np.float32, np.float64])
def test_writer_round_trip(tmp_path, no_h5py, dtype):
    rng = np.random.default_rng(0)
    arr = rng.normal(size=(4, 5, 6))
    if np.dtype(dtype).kind == 'c':
        arr = arr + 1j * rng.normal(size=arr.shape)
    arr = arr.astype(dtype)
    path = str(tmp_path / 'w.h5')
    h5io.write_dataset(path, 'exchange/data', arr)
    back = h5io.read_dataset(path)
    assert np.array_equal(back, arr)


def test_reader_rejects_garbage(tmp_path, no_h5py):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'not hdf5 at all' * 10)
    with pytest.raises(h5io.H5FormatError):
        h5io.read_dataset(str(p))


def test_writer_output_is_valid_for_h5py(tmp_path):
    """Cross-check with the real library when an interpreter that has it is around (build container)."""
    import subprocess
    conda = '/opt/conda/bin/python3.9'
    if not os.path.exists(conda):
        pytest.skip('no interpreter with h5py')
    rng = np.random.default_rng(1)
    arr = (rng.normal(size=(3, 4, 5)) + 1j * rng.normal(size=(3, 4, 5))).astype(np.complex64)
    path = str(tmp_path / 'w.h5')
    h5io.write_dataset(path, 'exchange/data', arr)
    np.save(str(tmp_path / 'w.npy'), arr)
    code = ("import h5py, numpy as np, sys; d = h5py.File(sys.argv[1], 'r')['exchange/data'][...]; "
            "assert d.dtype == np.complex64 and np.array_equal(d, np.load(sys.argv[2]))")
    subprocess.check_call([conda, '-c', code, path, str(tmp_path / 'w.npy')])


def test_tiff_round_trips(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.normal(size=(5, 7))
    vol = rng.normal(size=(3, 5, 7))
    f = tiffio.write_tiff(img, str(tmp_path / 'out' / 'img'), dtype='float32', overwrite=True)
    assert f.endswith('.tiff') and np.array_equal(tiffio.read_tiff(f), img.astype(np.float32))
    f = tiffio.write_tiff(vol, str(tmp_path / 'out' / 'vol'), dtype='float32', overwrite=True)
    assert np.array_equal(tiffio.read_tiff(f), vol.astype(np.float32))
    f2 = tiffio.write_tiff(img, str(tmp_path / 'out' / 'img'), dtype='float32', overwrite=False)
    assert f2 != str(tmp_path / 'out' / 'img.tiff')           # no clobbering without overwrite
    tiffio.write_tiff_stack(vol, str(tmp_path / 'fin_sup_mask' / 'mask'), dtype='float32', overwrite=True)
    back = tiffio.read_tiff_stack(str(tmp_path / 'fin_sup_mask' / 'mask_00000.tiff'), range(3), 5)
    assert np.array_equal(back, vol.astype(np.float32))


def test_create_noisy_data_fullfield_and_ptycho(tmp_path):
    """Poisson noise writer (tensorflow_recon/create_noisy_data.py:45-87): photon bookkeeping and statistics."""
    from beyond_dof_amd import h5io
    from beyond_dof_amd.simulation import create_noisy_data
    rng = np.random.RandomState(3)
    grid_delta = np.zeros((8, 8, 8))
    grid_delta[2:6, 2:6, 2:6] = 1e-6                                   # 64 sample voxels
    prj = (1.0 + 0.1 * rng.rand(4, 16, 16)).astype(np.complex64) * np.exp(0.3j)
    src = str(tmp_path / 'ff.h5')
    h5io.write_dataset(src, 'exchange/data', prj)
    dst = str(tmp_path / 'ff_noisy.h5')
    snr = create_noisy_data(src, dst, '6.4e5', grid_delta=grid_delta, rng=np.random.RandomState(1))
    out = np.asarray(h5io.read_dataset(dst))
    assert out.shape == prj.shape and out.dtype == np.complex64 and np.all(out.imag == 0)
    n_ph = 6.4e5 / 64                                                   # photons per unit intensity
    counts = np.abs(out) ** 2 * n_ph
    assert np.allclose(counts, np.round(counts), atol=1e-2)             # integer photon counts before the rescaling
    inten = np.abs(prj) ** 2
    z = (np.abs(out) ** 2 - inten) / np.sqrt(inten / n_ph)              # Poisson: variance = mean
    assert abs(z.mean()) < 0.1 and abs(z.std() - 1) < 0.1
    assert snr > 0
    # ptychography: photons per diffraction pattern = n_ph_tx * grid.size / n_sample / n_pos, whatever its intensity scale
    dp = (rng.rand(2, 5, 12, 12) * np.array([1, 10, 100, 1000, 1e4])[None, :, None, None]).astype(np.complex64)
    srcp = str(tmp_path / 'data_ptycho.h5')
    h5io.write_dataset(srcp, 'exchange/data', dp)
    dstp = str(tmp_path / 'data_ptycho_n.h5')
    create_noisy_data(srcp, dstp, 1e7, grid_delta=grid_delta, rng=np.random.RandomState(2))
    outp = np.asarray(h5io.read_dataset(dstp))
    n_ex = 1e7 * grid_delta.size / 64 / 5
    for i in range(2):
        for j in range(5):
            inten = np.abs(dp[i, j]) ** 2
            got = np.sum(np.abs(outp[i, j]) ** 2) * n_ex / np.sum(inten)
            assert abs(got / n_ex - 1) < 5 / np.sqrt(n_ex)              # total counts of the pattern ~ Poisson(n_ex)
    with pytest.raises(FileExistsError):
        create_noisy_data(src, dst, 1e6, grid_delta=grid_delta)


def test_bigtiff_round_trip(tmp_path):
    """Volumes beyond 4 GiB are written as BigTIFF (64-bit offsets); the format itself is exercised here on a small stack."""
    rng = np.random.RandomState(0)
    vol = rng.rand(3, 7, 5).astype(np.float32)
    f = tiffio.write_tiff(vol, str(tmp_path / 'big'), dtype='float32', overwrite=True, force_bigtiff=True)
    raw = open(f, 'rb').read()
    assert raw[:4] == b'II\x2b\x00'
    assert np.array_equal(tiffio.read_tiff(f), vol)
    f2 = tiffio.write_tiff(vol[0], str(tmp_path / 'big2'), dtype='float32', overwrite=True, force_bigtiff=True)
    assert np.array_equal(tiffio.read_tiff(f2), vol[0])
    try:
        import tifffile                         # only present in some images: an independent reader when available
        assert np.array_equal(tifffile.imread(f), vol)
    except ImportError:
        pass


def test_create_noisy_data_vs_the_reference_script(tmp_path, golden_dir):
    """Golden vector G16: tensorflow_recon/create_noisy_data.py run as the script it is (tests/golden/make_golden.py --g16;
    seed 1234 through its frozen clock, ptychography branch).  create_noisy_data, fed the same numpy stream in the script's
    order (three photon budgets x two output files), writes the same six datasets bit for bit."""
    import os
    from beyond_dof_amd import h5io
    from beyond_dof_amd.simulation import create_noisy_data
    g = np.load(os.path.join(golden_dir, 'g16_noisy_data.npz'))
    src = str(tmp_path / 'data_cell_phase_ptycho.h5')
    h5io.write_dataset(src, 'exchange/data', g['src'])
    rng = np.random.RandomState(int(g['seed']))
    for n_ph_tx in ('1.75e6', '1.75e7', '1.75e8'):
        for postfix in ('', '_ref'):
            dst = str(tmp_path / 'out_n{}{}.h5'.format(n_ph_tx, postfix))
            create_noisy_data(src, dst, n_ph_tx, grid_delta=g['grid_delta'], is_ptycho=True, rng=rng)
            want = g['data_cell_phase_n{}{}'.format(n_ph_tx, postfix).replace('.', 'p')]
            got = np.asarray(h5io.read_dataset(dst))
            assert got.dtype == np.complex64 and np.array_equal(got, want), (n_ph_tx, postfix)
