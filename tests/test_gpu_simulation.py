"""Dataset simulators (cnn_propagator/simulation.py): GPU forward vs the same recipe driven by the oracle."""
import os

import numpy as np
import pytest
from scipy.ndimage import rotate as sp_rotate

pytestmark = pytest.mark.gpu

from oracle import bdof_oracle as orc


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def test_create_fullfield_and_ptycho_data(tmp_path):
    import __graft_entry__ as entry
    entry.build()
    from beyond_dof_amd import h5io, simulation
    rng = np.random.default_rng(0)
    n = 64
    z, y, x = np.mgrid[:n, :n, :n]
    d = 2e-6 * np.exp(-((z - 30) ** 2 + (y - 34) ** 2 + (x - 28) ** 2) / (2 * 9.0 ** 2))
    ph = tmp_path / 'phantom'
    ph.mkdir()
    np.save(str(ph / 'grid_delta.npy'), d)
    np.save(str(ph / 'grid_beta.npy'), 0.1 * d)
    n_theta = 3
    dat = simulation.create_fullfield_data_numpy(5000., 1e-7, 1e-4, n_theta, str(ph), str(tmp_path), 'ff.h5', batch_size=2,
                                                 theta_end=2 * np.pi)
    back = h5io.read_dataset(str(tmp_path / 'ff.h5'))
    assert back.shape == (n_theta, n, n) and np.array_equal(back, dat)
    obj = np.stack([d, 0.1 * d], axis=-1)
    theta_ls = -np.linspace(0, 2 * np.pi, n_theta) / np.pi * 180
    for i, th in enumerate(theta_ls):
        rot = sp_rotate(obj, th, reshape=False, axes=(1, 2))[None]
        ref, _ = orc.multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], np.ones((n, n)), np.zeros((n, n)), 5000., 1e-7,
                                                      1e-4, rot[..., 0].shape, return_probe_array=False)
        assert rel(np.abs(dat[i]) ** 2, np.abs(ref[0]) ** 2) <= 1e-5
    with pytest.raises(FileExistsError):
        simulation.create_fullfield_data_numpy(5000., 1e-7, 1e-4, n_theta, str(ph), str(tmp_path), 'ff.h5')
    # ptychography: 64 x 64 probes on a 2 x 2 grid, no circular mask
    pos = [(20, 20), (20, 44), (44, 20), (44, 44)]
    pd = simulation.create_ptychography_data_batch_numpy(5000., 1e-7, 2, str(ph), str(tmp_path), 'pt.h5', pos, probe_size=(64, 64),
                                                         probe_circ_mask=None, minibatch_size=3, probe_mag_sigma=6.,
                                                         probe_phase_sigma=6., probe_phase_max=0.5, theta_end=np.pi)
    assert pd.shape == (2, 4, 64, 64)
    prr, pii = orc.gaussian_probe((64, 64), 6., 6., 0.5)
    rot = sp_rotate(obj, np.rad2deg(-np.pi), reshape=False, axes=(1, 2))
    pad, half = orc.ptycho_pad_amounts(np.array(pos), (64, 64), (n, n, n))
    obj_pad = np.pad(rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
    p = pos[3]
    sub = obj_pad[p[0] + pad[0, 0] - half[0]:p[0] + pad[0, 0] - half[0] + 64, p[1] + pad[1, 0] - half[1]:p[1] + pad[1, 0] - half[1] + 64][None]
    ref, _ = orc.multislice_propagate_batch_numpy(sub[..., 0], sub[..., 1], prr, pii, 5000., 1e-7, 'inf', sub[..., 0].shape,
                                                  return_probe_array=False)
    assert rel(np.abs(pd[1, 3]) ** 2, np.abs(ref[0]) ** 2) <= 1e-5
    assert os.path.exists(str(tmp_path / 'diffraction_dat' / 'mag_00001.tiff'))
