"""Dataset simulators — the writer side of the `exchange/data` contract (cnn_propagator/simulation.py:80-161,283-386):
same function names and arguments; the object is rotated on the host with scipy.ndimage.rotate (spline order 3,
reshape=False, axes=(1, 2), exactly the reference's call) and the multislice forward runs on the GPU.

Not ported: probe_type='point' (spherical-coordinate propagator, out of scope), the interactive overwrite prompt (an
existing file is overwritten only with overwrite=True).
"""
import os

import numpy as np
from scipy.ndimage import gaussian_filter
from scipy.ndimage import rotate as sp_rotate

from . import h5io, tiffio, util
from .np_funcs import multislice_propagate_batch_numpy

PI = util.PI


def circ_mask(arr, ratio=1.0, val=0.0):
    """Mask everything outside a centred circle of radius ratio * min(ny, nx) / 2 in the last two axes — what
    tomopy.circ_mask(arr, axis=0, ratio=ratio) does for a stack of images (tomopy itself is not a dependency)."""
    ny, nx = arr.shape[-2:]
    rad = ratio * min(ny, nx) / 2.0
    yy, xx = np.ogrid[0.5 - ny / 2.0:0.5 + ny / 2.0, 0.5 - nx / 2.0:0.5 + nx / 2.0]
    out = np.array(arr, dtype=float, copy=True)
    out[..., (yy * yy + xx * xx) >= rad * rad] = val
    return out


def _load_phantom(phantom_path):
    grid_delta = np.load(os.path.join(phantom_path, 'grid_delta.npy'))
    grid_beta = np.load(os.path.join(phantom_path, 'grid_beta.npy'))
    obj = np.zeros(np.append(grid_delta.shape, 2))
    obj[:, :, :, 0] = grid_delta
    obj[:, :, :, 1] = grid_beta
    return obj


def _check_target(path, overwrite):
    if os.path.exists(path) and not overwrite:
        raise FileExistsError('{} exists (pass overwrite=True)'.format(path))
    folder = os.path.dirname(path)
    if folder and not os.path.exists(folder):
        os.makedirs(folder)


def create_fullfield_data_numpy(energy_ev, psize_cm, free_prop_cm, n_theta, phantom_path, save_folder, fname, batch_size=1,
                                probe_type='plane', wavefront_initial=None, theta_st=0, theta_end=2 * PI,
                                monitor_output=False, overwrite=False, **kwargs):
    """cnn_propagator/simulation.py:80-161.  Writes (n_theta, Y, X) complex64 to save_folder/fname."""
    obj = _load_phantom(phantom_path)
    img_dim = obj.shape[:3]
    path = os.path.join(save_folder, fname)
    _check_target(path, overwrite)
    if probe_type == 'plane':
        probe_real, probe_imag = np.ones(img_dim[:2], dtype='float32'), np.zeros(img_dim[:2], dtype='float32')
    elif probe_type == 'fixed':
        probe_real, probe_imag = util.mag_phase_to_real_imag(*wavefront_initial)
    elif probe_type == 'gaussian':
        probe_real, probe_imag = util.gaussian_probe(img_dim[:2], kwargs['probe_mag_sigma'], kwargs['probe_phase_sigma'],
                                                     kwargs['probe_phase_max'])
    else:
        raise ValueError("Invalid wavefront type. Choose from 'plane', 'fixed' or 'gaussian'.")
    theta_ls = -np.linspace(theta_st, theta_end, n_theta) / np.pi * 180
    theta_batch = np.array_split(theta_ls, int(np.ceil(float(n_theta) / batch_size)))
    dat = np.zeros((n_theta, img_dim[0], img_dim[1]), dtype=np.complex64)
    pos = 0
    for i_batch, this_theta_batch in enumerate(theta_batch):
        rot = np.array([sp_rotate(obj, theta, reshape=False, axes=(1, 2)) for theta in this_theta_batch])
        wave_out, _ = multislice_propagate_batch_numpy(rot[..., 0], rot[..., 1], probe_real, probe_imag, energy_ev, psize_cm,
                                                       free_prop_cm=free_prop_cm, obj_batch_shape=rot.shape[:-1],
                                                       return_probe_array=False)
        if monitor_output:
            tiffio.write_tiff(np.abs(wave_out), os.path.join(save_folder, 'monitor_output', 'prj_{}'.format(i_batch)),
                              dtype='float32', overwrite=True)
        dat[pos:pos + len(this_theta_batch)] = wave_out
        pos += len(this_theta_batch)
    h5io.write_dataset(path, 'exchange/data', dat)
    return dat


def create_ptychography_data_batch_numpy(energy_ev, psize_cm, n_theta, phantom_path, save_folder, fname, probe_pos,
                                         probe_type='gaussian', probe_size=(72, 72), wavefront_initial=None,
                                         theta_st=0, theta_end=2 * PI, probe_circ_mask=0.9, minibatch_size=20,
                                         overwrite=False, **kwargs):
    """cnn_propagator/simulation.py:283-386.  Writes (n_theta, n_pos, py, px) complex64 far-field waves."""
    obj = _load_phantom(phantom_path)
    img_dim = obj.shape[:3]
    probe_pos = np.array(probe_pos)
    n_pos = len(probe_pos)
    minibatch_size = min([minibatch_size, n_pos])
    probe_pos_batches = np.array_split(probe_pos, int(np.ceil(float(n_pos) / minibatch_size)))
    half = (np.array(probe_size) / 2).astype('int')
    path = os.path.join(save_folder, fname)
    _check_target(path, overwrite)
    if probe_type != 'gaussian':
        raise ValueError("only probe_type='gaussian' is generated here, as in the reference")
    probe_real, probe_imag = util.gaussian_probe(probe_size, kwargs['probe_mag_sigma'], kwargs['probe_phase_sigma'],
                                                 kwargs['probe_phase_max'])
    probe_mask = None
    if probe_circ_mask is not None:
        probe_real, probe_imag = circ_mask(np.array([probe_real, probe_imag]), ratio=probe_circ_mask)
        probe_mask = gaussian_filter(np.squeeze(circ_mask(np.ones((1,) + tuple(probe_size)), ratio=probe_circ_mask)), 3)
    pad = np.array([[0, 0], [0, 0]])
    if probe_pos[:, 0].min() - half[0] < 0:
        pad[0, 0] = half[0] - probe_pos[:, 0].min()
    if probe_pos[:, 0].max() + half[0] > img_dim[0]:
        pad[0, 1] = probe_pos[:, 0].max() + half[0] - img_dim[0]
    if probe_pos[:, 1].min() - half[1] < 0:
        pad[1, 0] = half[1] - probe_pos[:, 1].min()
    if probe_pos[:, 1].max() + half[1] > img_dim[1]:
        pad[1, 1] = probe_pos[:, 1].max() + half[1] - img_dim[1]      # (the reference slips to half[0] here, quirk Q14)
    theta_ls = np.rad2deg(-np.linspace(theta_st, theta_end, n_theta))
    dat = np.zeros((n_theta, n_pos, probe_size[0], probe_size[1]), dtype=np.complex64)
    for ii, theta in enumerate(theta_ls):
        obj_rot = sp_rotate(obj, theta, reshape=False, axes=(1, 2))
        obj_rot = np.pad(obj_rot, ((pad[0, 0], pad[0, 1]), (pad[1, 0], pad[1, 1]), (0, 0), (0, 0)), mode='constant')
        out = []
        for pos_batch in probe_pos_batches:
            subs = np.array([obj_rot[int(p[0]) + pad[0, 0] - half[0]:int(p[0]) + pad[0, 0] - half[0] + probe_size[0],
                                     int(p[1]) + pad[1, 0] - half[1]:int(p[1]) + pad[1, 0] - half[1] + probe_size[1]]
                             for p in pos_batch])
            exiting, _ = multislice_propagate_batch_numpy(subs[..., 0], subs[..., 1], probe_real, probe_imag, energy_ev, psize_cm,
                                                          free_prop_cm='inf', obj_batch_shape=subs.shape[:-1],
                                                          return_probe_array=False)
            if probe_mask is not None:
                exiting = exiting * probe_mask
            out.append(exiting)
        dat[ii] = np.vstack(out)
        tiffio.write_tiff(np.abs(dat[ii]), os.path.join(save_folder, 'diffraction_dat', 'mag_{:05d}'.format(ii)),
                          overwrite=True, dtype='float32')
    h5io.write_dataset(path, 'exchange/data', dat)
    return dat


def create_noisy_data(src_fname, dest_fname, n_ph_tx, grid_delta=None, n_sample_pixel=None, is_ptycho=None, rng=None,
                      overwrite=False):
    """Photon (Poisson) noise on a simulated dataset — tensorflow_recon/create_noisy_data.py:20-87 as a function.

    n_ph_tx: total number of photons through the sample (the script's '1.75e6' ... strings); n_sample_pixel: voxels of
    the phantom with delta > 1e-10 (:14, from grid_delta when not given).  Full field (:74-86): every projection's
    intensity |prj|^2 is scaled to n_ph = n_ph_tx / n_sample_pixel photons per unit intensity, Poisson-sampled, scaled
    back, and stored as its square root.  Ptychography (:45-72): the photons per diffraction pattern are
    n_ph_tx * grid_delta.size / n_sample_pixel / n_positions, distributed over the pattern in proportion to its intensity.
    The noisy AMPLITUDE goes to `exchange/data` as complex64 (phase is lost, as in the reference).  Returns the mean
    signal-to-noise ratio var(I) / var(noise) the script prints (:88)."""
    rng = rng or np.random
    src = h5io.File(src_fname)['exchange/data']
    if is_ptycho is None:
        is_ptycho = 'ptycho' in src_fname or len(src.shape) == 4              # :29-31
    if n_sample_pixel is None:
        if grid_delta is None:
            raise ValueError('give grid_delta (the phantom) or n_sample_pixel')
        n_sample_pixel = int(np.count_nonzero(np.asarray(grid_delta) > 1e-10))
    n_ph = float(n_ph_tx) / n_sample_pixel
    _check_target(dest_fname, overwrite)
    out = np.empty(src.shape, dtype=np.complex64)
    snr = []
    if is_ptycho:
        if grid_delta is None:
            raise ValueError('the ptychography branch needs grid_delta (photons per image scale with its size, :49-51)')
        n_ex = n_ph * n_sample_pixel * (float(np.asarray(grid_delta).size) / n_sample_pixel) / src.shape[1]
        for i in range(src.shape[0]):
            block = np.asarray(src[i])
            for j in range(src.shape[1]):
                inten = np.abs(block[j]) ** 2
                multiplier = n_ex / np.sum(inten)
                noisy = rng.poisson(inten * multiplier) / multiplier
                snr.append(np.var(inten) / np.var(noisy - inten))
                out[i, j] = np.sqrt(noisy).astype(np.complex64)
    else:
        for i in range(src.shape[0]):
            inten = np.abs(np.asarray(src[i])) ** 2
            noisy = rng.poisson(inten * n_ph) / n_ph
            snr.append(np.var(inten) / np.var(noisy - inten))
            out[i] = np.sqrt(noisy).astype(np.complex64)
    h5io.write_dataset(dest_fname, 'exchange/data', out)
    return float(np.mean(snr))
