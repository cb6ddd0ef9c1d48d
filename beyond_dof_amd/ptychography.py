"""reconstruct_ptychography — drop-in for cnn_propagator/ptychography.py:19-365 with the forward + gradient loop on the
GPU.  Same keyword surface (unknown keywords ignored), `exchange/data` HDF5 of shape (n_theta, n_pos, py, px) read
lazily per minibatch, same outputs.

Differences: by default the FFT propagator instead of the truncated real-space convolution
(propagator='conv' selects the latter for power-of-two probes); small square probes (32 ... 128 pixels, the reference
drivers' 72 x 72 among them) run on the LDS-resident kernel, powers of two from 64 to 1024 on the fused streaming kernels, any
other size on the rocFFT engine; probe_type='optimizable' optimises the probe as the TF twin does; Q9 dynamic dropping is a
no-op in the reference and is not run; the intermediate TIFF is behind save_intermediate; n_epochs='auto' stops at max_nepochs.
"""
import os
import time

import numpy as np

from . import h5io, tiffio, util
from .comm import get_comm
from .fullfield import create_probe_initial_guess, upsample_2x
from .misc import create_summary
from .solver import PtychoSolver
from ._lib import BdofError
from .util import print_flush, split_tasks

PI = util.PI


def epoch_schedule(n_theta, n_pos, minibatch_size, rng):
    """(theta, position) list of one epoch, cnn_propagator/ptychography.py:269-282: thetas shuffled, each theta's position
    list padded to a multiple of minibatch_size with a random choice of its first positions (quirk Q10)."""
    theta_ls = np.arange(n_theta)
    rng.shuffle(theta_ls)
    rows = []
    for i_theta in theta_ls:
        spots = np.arange(n_pos)
        if n_pos % minibatch_size != 0:
            spots = np.append(spots, rng.choice(spots[:-(n_pos % minibatch_size)], minibatch_size - (n_pos % minibatch_size),
                                                replace=False))
        rows.append(np.stack([np.full(len(spots), i_theta), spots], axis=1))
    return np.concatenate(rows, axis=0)


def batches_of_epoch(n_theta, n_pos, minibatch_size, size, rank, rng):
    """[(i_theta, sorted position indices of this rank)] of one epoch, cnn_propagator/ptychography.py:253-297: the epoch's
    (theta, position) list in chunks of size * minibatch_size; a short chunk is topped up from the first one; the NUMBER of
    chunks that are run is ceil(n_theta * n_pos / (size * minibatch_size)) — counted from the un-padded number of spots
    (:257-260), so when n_pos is not a multiple of the minibatch the tail of the padded list is never visited; the angle is
    the one of the rank's first entry.  Pinned by golden vector G11 (the reference's own loop executed)."""
    n_tot = minibatch_size * size
    chunks = split_tasks(epoch_schedule(n_theta, n_pos, minibatch_size, rng), n_tot)
    n_batch = int(np.ceil(float(n_theta * n_pos) / n_tot))
    out = []
    for i_batch in range(n_batch):
        chunk = chunks[i_batch]
        if len(chunk) < n_tot:
            chunk = np.concatenate([chunk, chunks[0][:n_tot - len(chunk)]])
        mine = chunk[rank * minibatch_size:(rank + 1) * minibatch_size]
        out.append((int(chunk[rank * minibatch_size, 0]), np.sort(mine[:, 1])))
    return out


def reconstruct_ptychography(fname, probe_pos, probe_size, obj_size, theta_st=0, theta_end=PI, theta_downsample=None,
                             n_epochs='auto', crit_conv_rate=0.03, max_nepochs=200,
                             alpha=1e-7, alpha_d=None, alpha_b=None, gamma=1e-6, learning_rate=1.0,
                             output_folder=None, minibatch_size=None, save_intermediate=False, full_intermediate=False,
                             energy_ev=5000, psize_cm=1e-7, cpu_only=False, save_path='.',
                             phantom_path='phantom', core_parallelization=True, free_prop_cm=None,
                             multiscale_level=1, n_epoch_final_pass=None, initial_guess=None, n_batch_per_update=1,
                             dynamic_rate=True, probe_type='gaussian', probe_initial=None, probe_learning_rate=1e-3,
                             pupil_function=None, probe_circ_mask=0.9, finite_support_mask=None,
                             forward_algorithm='fresnel', dynamic_dropping=True, dropping_threshold=8e-5,
                             n_dp_batch=20, object_type='normal', **kwargs):
    t_zero = time.time()
    comm = kwargs.get('comm') or get_comm()
    size, rank = comm.size, comm.rank
    seed = kwargs.get('seed', int(time.time() / 60))
    variant = kwargs.get('variant', 'numpy_skip_last')
    # 'fft' (default): transfer-function propagator of np_funcs.py; 'conv': truncated real-space kernel of propagation.py,
    # the reference's own choice in this entry point, with `kernel_size` taps per axis
    propagator = kwargs.get('propagator', 'fft')
    # 'float64': the adjoint sweep in float64 (bdof_configure flag 64; with propagator='conv' both sweeps, bdof_loss_grad_conv_f64);
    # 'first-step': the first minibatch of every epoch through the model's float64 path on the same context (bdof_loss_grad_tf_f64 /
    # bdof_loss_grad_conv_f64: both sweeps in double, no second engine) — Adam's first step after its per-epoch restart is
    # lr g / (|g| + 1e-8), the one step in which the float32 rounding of the gradient reaches the volume
    # (the default with the transfer-function propagator: reconstructed delta within 2.1e-6 of the reference's float64 loop on golden
    # vector G17, 7.0e-6 with 'float32' — both inside 1e-5 since the kernels take dithered copies of the transfer function;
    # one slower step per epoch)
    adjoint_precision = kwargs.get('adjoint_precision', 'first-step')
    if adjoint_precision not in ('float32', 'float64', 'first-step'):
        raise ValueError("adjoint_precision must be 'float32', 'float64' or 'first-step'")

    print_flush('Reading data...', 0, rank)
    f = h5io.File(os.path.join(save_path, fname))
    prj = f['exchange/data']                               # kept open, read per minibatch (ptychography.py:91-92,295)
    n_theta = prj.shape[0]
    prj_theta_ind = np.arange(n_theta, dtype=int)
    if theta_downsample is not None:
        prj_theta_ind = prj_theta_ind[::theta_downsample]
        n_theta = len(prj_theta_ind)
    original_shape = [n_theta] + list(prj.shape[1:])
    print_flush('Data shape: {}'.format(original_shape), 0, rank)

    if output_folder is None:
        output_folder = 'recon_ptycho_minibatch_{}_iter_{}_alphad_{}_alphab_{}_rate_{}_energy_{}_size_{}_ntheta_{}_ms_{}_cpu_{}'.format(
            minibatch_size, n_epochs, alpha_d, alpha_b, learning_rate, energy_ev, prj.shape[-1], prj.shape[0], multiscale_level, cpu_only)
        if abs(PI - theta_end) < 1e-3:
            output_folder += '_180'
    if save_path != '.':
        output_folder = os.path.join(save_path, output_folder)

    probe_pos = np.array(probe_pos)
    n_pos = len(probe_pos)
    obj_delta = obj_beta = None
    first_level = True
    for ds_level in range(multiscale_level - 1, -1, -1):
        ds_level = 2 ** ds_level
        print_flush('Multiscale downsampling level: {}'.format(ds_level), 0, rank)
        this_obj_size = [int(x / ds_level) for x in obj_size] if ds_level > 1 else list(obj_size)
        this_probe_size = [int(x / ds_level) for x in probe_size] if ds_level > 1 else list(probe_size)
        this_probe_pos = (probe_pos / ds_level).astype(int) if ds_level > 1 else probe_pos
        if minibatch_size is None:
            minibatch_size = n_pos

        coord_ls = util.rotation_lookup_files(this_obj_size, n_theta, comm)

        np.random.seed(seed)                               # rank 0 initialises and broadcasts in the reference (:169-208)
        if first_level:
            if initial_guess is None:
                obj_delta = np.random.normal(size=this_obj_size, loc=8.7e-7, scale=1e-7)
                obj_beta = np.random.normal(size=this_obj_size, loc=5.1e-8, scale=1e-8)
            else:
                obj_delta, obj_beta = np.array(initial_guess[0], dtype=float), np.array(initial_guess[1], dtype=float)
        else:
            obj_delta = upsample_2x(obj_delta) + np.random.normal(size=this_obj_size, loc=8.7e-7, scale=1e-7)
            obj_beta = upsample_2x(obj_beta) + np.random.normal(size=this_obj_size, loc=5.1e-8, scale=1e-8)
        obj_delta[obj_delta < 0] = 0
        obj_beta[obj_beta < 0] = 0
        if object_type == 'phase_only':
            obj_beta[...] = 0
        elif object_type == 'absorption_only':
            obj_delta[...] = 0
        if size > 1:
            obj_delta, obj_beta = comm.bcast_host(obj_delta, 0), comm.bcast_host(obj_beta, 0)

        if probe_type == 'gaussian':
            probe_mag_sigma, probe_phase_sigma, probe_phase_max = (kwargs['probe_mag_sigma'], kwargs['probe_phase_sigma'],
                                                                  kwargs['probe_phase_max'])
            probe_real, probe_imag = util.gaussian_probe(this_probe_size, probe_mag_sigma, probe_phase_sigma, probe_phase_max)
        elif probe_type == 'optimizable':
            if probe_initial is not None:
                probe_real, probe_imag = util.mag_phase_to_real_imag(*probe_initial)
            else:
                back_prop_cm = (free_prop_cm + psize_cm * obj_delta.shape[2]) if free_prop_cm is not None else psize_cm * obj_delta.shape[2]
                probe_init = create_probe_initial_guess(os.path.join(save_path, fname), back_prop_cm * 1.e7, energy_ev, psize_cm * 1.e7)
                probe_real, probe_imag = probe_init.real, probe_init.imag
            if pupil_function is not None:
                probe_real, probe_imag = probe_real * pupil_function, probe_imag * pupil_function
        elif probe_type == 'fixed':
            probe_real, probe_imag = util.mag_phase_to_real_imag(*probe_initial)
        else:
            raise ValueError("Invalid wavefront type. Choose from 'plane', 'fixed', 'optimizable'.")

        mk = lambda prec: PtychoSolver(this_obj_size, this_probe_size, this_probe_pos, n_theta, minibatch_size, energy_ev,
                                       psize_cm * ds_level, probe_real, probe_imag, variant=variant, comm=comm, device=comm.local_rank,
                                       coord_ls=coord_ls, propagator=propagator, kernel_size=kwargs.get('kernel_size', 17),
                                       adjoint64={'float32': None, 'float64': True, 'first-step': 'first'}[prec])
        # the DEFAULT's float64 path (first minibatch of an epoch) is an accuracy refinement with buffers of its own (wave + tape in
        # complex128, allocated when the solver is built): where they do not fit beside the engine's the run goes on in float32,
        # SAYS so and records it (summary.txt: adjoint_precision_effective); an explicit request fails.  The retry runs outside the
        # except block and after a collection: the exception's traceback holds the frame of PtychoSolver.__init__ — the engine, its
        # tape and the partly allocated float64 buffers — and would keep them on the device while the fallback allocates.
        adjoint_precision_effective, setup_error = adjoint_precision, None
        try:
            solver = mk(adjoint_precision)
        except BdofError as err:
            if 'adjoint_precision' in kwargs or adjoint_precision != 'first-step':
                raise
            setup_error = str(err)
        if setup_error is not None:
            import gc
            gc.collect()
            adjoint_precision_effective = 'float32'
            print_flush("adjoint_precision='first-step' could not be set up ({}): continuing with 'float32' — the reconstructed "
                        "delta is then within ~7e-6 of a float64 run instead of ~2e-6 (golden vector G17; DESIGN.md, numerics)".format(
                            setup_error), 0, rank)
            solver = mk('float32')
        solver.set_volume(obj_delta, obj_beta)
        solver.tune_tail()
        # the diffraction amplitudes stay on the device when they fit (ptychography.py:295 reads them from the file per step)
        resident = int(np.prod(prj.shape)) * 4 <= int(kwargs.get('resident_data_bytes', 16 << 30))
        if resident:
            dat = np.abs(np.asarray(prj[...]))
            dat = dat[prj_theta_ind]
            if ds_level > 1:
                dat = dat[:, :, ::ds_level, ::ds_level]
            solver.set_measurements(dat)
            del dat
        if probe_type == 'optimizable':          # tensorflow_recon/ptychography.py: the probe is a variable with its own Adam
            solver.enable_probe_optimization(probe_real, probe_imag, probe_learning_rate, pupil_function)
        print_flush('Optimizer started.', 0, rank)
        if rank == 0:
            create_summary(output_folder, locals(), preset='ptycho')

        rng = np.random.RandomState(seed)
        i_epoch, cont = 0, True
        while cont:
            t0 = time.time()
            batches = batches_of_epoch(n_theta, n_pos, minibatch_size, size, rank, rng)
            solver.reset_moments()                          # m, v = (None, None), ptychography.py:262
            for i_batch, (this_i_theta, this_ind_rank) in enumerate(batches):
                this_prj_batch = None
                if not resident:
                    this_prj_batch = np.abs(prj[int(prj_theta_ind[this_i_theta]), this_ind_rank.tolist()])
                    if ds_level > 1:
                        this_prj_batch = this_prj_batch[:, ::ds_level, ::ds_level]
                # loss_grad -> Allreduce -> /size -> Adam -> clip (ptychography.py:301-310): one pipelined device step
                solver.step(i_batch, this_i_theta, this_ind_rank, this_prj_batch, learning_rate)
                if save_intermediate and rank == 0:
                    d, _ = solver.get_volume()
                    tiffio.write_tiff(d, os.path.join(output_folder, 'intermediate', 'current'), dtype='float32', overwrite=True)
            i_epoch += 1
            this_loss = solver.loss_and_grad(this_i_theta, this_ind_rank, this_prj_batch, want_loss=True)
            print_flush('Epoch {} (rank {}); loss = {}; Delta-t = {} s; current time = {} s,'.format(
                i_epoch, rank, this_loss, time.time() - t0, time.time() - t_zero), 0, rank)
            if n_epochs == 'auto':
                cont = i_epoch < max_nepochs
            else:
                cont = i_epoch < n_epochs
            obj_delta, obj_beta = solver.get_volume()
            if rank == 0:
                tiffio.write_tiff(obj_delta, os.path.join(output_folder, 'delta_ds_{}'.format(ds_level)), dtype='float32', overwrite=True)
                tiffio.write_tiff(obj_beta, os.path.join(output_folder, 'beta_ds_{}'.format(ds_level)), dtype='float32', overwrite=True)
        obj_delta, obj_beta = obj_delta.astype(float), obj_beta.astype(float)
        first_level = False
        del solver
        print_flush('Current iteration finished.', 0, rank)
    comm.Barrier()
    return obj_delta, obj_beta
