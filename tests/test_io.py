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
