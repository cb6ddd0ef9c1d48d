"""Host-side set-up code of the product (beyond_dof_amd/util.py, comm.py) against the golden vectors
captured from the reference and against the oracle.  No GPU needed."""
import os

import numpy as np

from beyond_dof_amd import util
from beyond_dof_amd.comm import minibatch_schedule, PseudoComm
from oracle import bdof_oracle as orc


def test_get_kernel_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g1_get_kernel.npz'))
    for key in g.files:
        _, Y, X, dist = key.split('_')
        H = util.get_kernel(float(dist), 0.248, [1., 1., 1.], [int(Y), int(X), 4])
        np.testing.assert_allclose(H, g[key], rtol=0, atol=1e-13)


def test_get_kernel_anisotropic_voxels_match_oracle():
    H = util.get_kernel(3.0, 0.3, [1.5, 2.5, 1.0], [10, 12])
    np.testing.assert_allclose(H, orc.get_kernel(3.0, 0.3, [1.5, 2.5, 1.0], [10, 12, 1]), rtol=0, atol=1e-14)


def test_device_transfer_function_is_the_fft_domain_multiplier():
    ny, nx = 8, 16
    hs = util.device_transfer_function(1.0, 0.248, [1., 1., 1.], ny, nx)
    rng = np.random.default_rng(0)
    w = rng.normal(size=(1, ny, nx)) + 1j * rng.normal(size=(1, ny, nx))
    ref = orc._propagate(w, orc.get_kernel(1.0, 0.248, [1., 1., 1.], (ny, nx, 1)))
    mine = np.fft.fft2(w[0]) * (hs * nx * ny)
    mine = np.fft.ifft2(mine)
    np.testing.assert_allclose(mine, ref[0], rtol=0, atol=5e-7)


def test_rotation_lookup_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g3_rotation.npz'))
    for size, n in [((8, 8, 8), 5), ((64, 64, 64), 4), ((6, 10, 10), 7)]:
        key = 'x'.join(map(str, size)) + '_n{}'.format(n)
        assert np.array_equal(np.stack(util.rotation_lookup(list(size), n)), g['coords_' + key])


def test_save_rotation_lookup_files(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g3_rotation.npz'))
    folder = str(tmp_path / 'tables')
    util.save_rotation_lookup([8, 8, 8], 5, dest_folder=folder)
    coords = util.read_all_origin_coords(folder, 5)
    assert np.array_equal(np.stack(coords), g['coords_8x8x8_n5'])
    c0, c1, c2 = [np.load(os.path.join(folder, 'coord{}_vec.npy'.format(i))) for i in range(3)]
    # the flat (i0, i1, i2) enumeration apply_rotation relies on (cnn_propagator/util.py:386-396)
    assert np.array_equal(c0, np.repeat(np.arange(8), 64))
    assert np.array_equal(c1, np.tile(np.repeat(np.arange(8), 8), 8))
    assert np.array_equal(c2, np.tile(np.arange(8), 64))


def test_device_rotation_tables_gather_and_inverse():
    ny, nx, nz, n_theta = 4, 9, 9, 6
    coords = util.rotation_lookup([ny, nx, nz], n_theta)
    tab, off, order = util.device_rotation_tables(coords, nx, nz)
    rng = np.random.default_rng(0)
    obj = rng.normal(size=(ny, nx, nz, 2))
    rows = util.volume_to_rows(obj[..., 0], obj[..., 1]).reshape(nx * nz, ny, 2)
    for a in range(n_theta):
        rot = orc.apply_rotation(obj, coords[a])                    # (Y, X, Z, 2)
        mine = rows[tab[a]]                                         # [z][x][y][2]
        np.testing.assert_array_equal(mine.transpose(2, 1, 0, 3), rot.astype(np.float32))
        # inverse table: every destination lists exactly the rotated rows gathered from it
        dest = tab[a].reshape(-1)
        for d in range(nx * nz):
            src = order[a][off[a][d]:off[a][d + 1]]
            assert np.array_equal(np.sort(src), np.flatnonzero(dest == d))
        assert off[a][-1] == nx * nz


def test_layout_round_trips():
    rng = np.random.default_rng(1)
    d, b = rng.normal(size=(3, 4, 5)), rng.normal(size=(3, 4, 5))
    d2, b2 = util.rows_to_volume(util.volume_to_rows(d, b))
    np.testing.assert_array_equal(d2, d.astype(np.float32))
    np.testing.assert_array_equal(b2, b.astype(np.float32))
    gd, gb = rng.normal(size=(2, 3, 4, 5)), rng.normal(size=(2, 3, 4, 5))
    r = util.batch_to_rows(gd, gb)
    assert r.shape == (2, 5, 4, 3, 2)
    gd2, gb2 = util.rows_to_batch(r)
    np.testing.assert_array_equal(gd2, gd.astype(np.float32))


def test_split_tasks_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g6_split_tasks.npz'))
    parts = util.split_tasks(g['arr'], int(g['split_size']))
    assert [len(p) for p in parts] == list(g['lengths'])


def test_minibatch_schedule_pads_and_partitions():
    rng = np.random.default_rng(0)
    sched = minibatch_schedule(23, size=2, minibatch_size=4, rng=rng)
    assert all(len(c) == 8 for c in sched) and len(sched) == 3
    assert all(np.all(np.diff(c) >= 0) for c in sched)
    flat = np.concatenate(sched)
    assert set(flat.tolist()) == set(range(23))          # every angle is visited, one is repeated as padding
    assert len(flat) == 24
    assert PseudoComm().size == 1


def test_gaussian_probe_matches_oracle():
    a = util.gaussian_probe((7, 9), 2.0, 3.0, 0.5)
    b = orc.gaussian_probe((7, 9), 2.0, 3.0, 0.5)
    np.testing.assert_allclose(a[0], b[0])
    np.testing.assert_allclose(a[1], b[1])


def test_conv_kernel_separable_matches_reference_recipe():
    """The separable taps reproduce the reference's 2-D kernel (propagation.py:35-44) restated in the oracle."""
    for shape, ks in [((64, 64, 8), 17), ((64, 128, 8), 5), ((96, 64, 4), 9)]:
        ky, kx, e = util.conv_kernel_separable(1.0, 0.248, np.array([1., 1., 1.]), shape, ks)
        k2 = orc.conv_kernel_2d(1.0, 0.248, np.array([1., 1., 1.]), np.array(shape), ks)
        np.testing.assert_allclose(e * np.outer(ky, kx), k2, rtol=0, atol=1e-13 * np.abs(k2).max())


def test_ptychography_epoch_schedule():
    """cnn_propagator/ptychography.py:269-282: every angle once, in shuffled order; each angle's position list padded to a
    multiple of the minibatch with positions drawn from its own head, so that minibatches never mix angles."""
    from beyond_dof_amd.ptychography import epoch_schedule
    n_theta, n_pos, mb = 5, 23, 4
    sched = epoch_schedule(n_theta, n_pos, mb, np.random.RandomState(3))
    per_theta = 24
    assert sched.shape == (n_theta * per_theta, 2)
    assert sorted(set(sched[:, 0].tolist())) == list(range(n_theta))
    for i in range(n_theta):
        blk = sched[i * per_theta:(i + 1) * per_theta]
        assert len(set(blk[:, 0].tolist())) == 1                       # one angle per block -> per minibatch
        assert np.array_equal(blk[:n_pos, 1], np.arange(n_pos))
        assert np.all(blk[n_pos:, 1] < n_pos - n_pos % mb)             # padding drawn from spots[:-(n_pos % mb)]
    again = epoch_schedule(n_theta, n_pos, mb, np.random.RandomState(3))
    assert np.array_equal(sched, again)                                # every rank derives the same schedule from the seed
    exact = epoch_schedule(2, 8, 4, np.random.RandomState(0))
    assert exact.shape == (16, 2)


def test_rotation_tables_square_and_non_square():
    """util.rotation_lookup equals the reference's tables (oracle restatement, golden vector G3) whenever the rotation plane is
    square (X = Z) — every case the reference drivers run.  For X != Z the reference enumerates the new coordinates with
    np.reshape(np.tile(coord1, Z), [X, Z]) (cnn_propagator/util.py:304-306), which is only the intended repeat when X = Z:
    its (x, z) pairs then repeat some voxels and skip others, i.e. the rotated object is scrambled.  The product keeps the
    intended enumeration there — a deliberate deviation, found by executing the reference's loop on a (64, 64, 32) object."""
    from oracle import bdof_oracle as orc
    for size in ([8, 8, 8], [6, 10, 10], [12, 5, 5]):
        assert all(np.array_equal(a, b) for a, b in zip(orc.rotation_lookup(size, 5), util.rotation_lookup(size, 5)))
    size = [8, 10, 6]
    ours = util.rotation_lookup(size, 1)[0]                       # theta = 0: the identity map of the enumeration
    pairs = set(map(tuple, ours))
    assert len(pairs) == size[1] * size[2]                        # every (x, z) once
    theirs = orc.rotation_lookup(size, 1)[0]
    assert len(set(map(tuple, theirs))) < size[1] * size[2]       # the reference's enumeration repeats pairs


def test_point_probe_is_refused_by_both_entry_points(tmp_path):
    """probe_type='point' needs the spherical-wave propagator (cnn_propagator/fullfield.py:298-301 says so itself): refused by
    reconstruct_fullfield as by the simulators, before anything touches a GPU."""
    import pytest
    from beyond_dof_amd import h5io
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    h5io.write_dataset(str(tmp_path / 'data.h5'), 'exchange/data', np.ones((2, 8, 8), dtype=np.complex64))
    with pytest.raises(ValueError, match='point'):
        reconstruct_fullfield('data.h5', save_path=str(tmp_path), n_epochs=1, minibatch_size=1, probe_type='point',
                              initial_guess=[np.zeros((8, 8, 8)), np.zeros((8, 8, 8))])


def test_adjoint_precision_values_are_checked_before_anything_touches_a_gpu(tmp_path):
    """adjoint_precision: 'float32' | 'float64' | 'first-step' in both entry points (the first minibatch of an epoch through the
    model's float64 path); anything else is a ValueError at the top of the call."""
    import pytest
    from beyond_dof_amd.fullfield import reconstruct_fullfield
    from beyond_dof_amd.ptychography import reconstruct_ptychography
    with pytest.raises(ValueError, match='adjoint_precision'):
        reconstruct_fullfield('data.h5', save_path=str(tmp_path), n_epochs=1, minibatch_size=1, adjoint_precision='double')
    with pytest.raises(ValueError, match='adjoint_precision'):
        reconstruct_ptychography('data.h5', [(4, 4)], (4, 4), (8, 8, 8), save_path=str(tmp_path), n_epochs=1, minibatch_size=1,
                                 adjoint_precision='first')


def test_tiled_propagator_auto_halo():
    """TiledPropagator(halo='auto') (host arithmetic only): 64 pixels for plain stitching; with the long-range correction twice the
    band edge's reach over one 16-slice range plus 16 pixels — 24 at 5 keV / 1 nm (81 tiles of 512^2 on the 4096^2 field) — or plus 40
    where the tiles ride on per-tile carriers (48: 100 tiles, of which the few that touch the object run)."""
    from beyond_dof_amd.tiling import TiledPropagator

    class Probe(TiledPropagator):
        def __init__(self, n_slice, tile):
            self.n_slice, self.tile = n_slice, tile
    assert Probe(1024, 512)._auto_halo(5000., 1e-7, 0.5, None, 'auto', None) == 24           # full-amplitude float32 sweeps (gradient)
    assert Probe(1024, 512)._auto_halo(5000., 1e-7, 0.5, None, 'auto', None, True) == 48     # per-tile carriers (the forward model)
    assert Probe(1024, 512)._auto_halo(5000., 1e-7, 0.5, None, False, None) == 64
    assert Probe(96, 512)._auto_halo(5000., 1e-7, 0.5, None, 'auto', None, True) == 64       # one plain range: no correction
    assert Probe(1024, 128)._auto_halo(5000., 1e-7, 0.5, None, 'auto', None) == 24
    assert Probe(1024, 64)._auto_halo(5000., 1e-7, 0.5, None, True, None, True) == 16        # never more than a quarter of the tile
    assert (-(-4096 // (512 - 2 * 24))) ** 2 == 81 and (-(-4096 // (512 - 2 * 48))) ** 2 == 100


def test_detector_kernel_choice_and_impulse_response_table(golden_dir):
    """SURVEY §8 a3 on the host: 'TF' / 'IR' / 'auto' for the detector step (cnn_propagator/np_funcs.py:51-61: the criterion is
    computed there and then overridden with 'TF'); the device table of the 'IR' choice is get_kernel_ir (golden vector G8) in
    the layout of every other table; the oracle's 'IR' branch multiplies by that kernel."""
    import os
    import pytest
    from oracle import bdof_oracle as orc
    assert util.detector_kernel_kind('TF', 1e3, 0.248, [1., 1., 1.], (64, 64)) == 'TF'
    assert util.detector_kernel_kind('IR', 1.0, 0.248, [1., 1., 1.], (64, 64)) == 'IR'
    assert util.detector_kernel_kind('auto', 1e3, 0.248, [1., 1., 1.], (64, 64)) == 'IR'      # lambda z / L = 3.9 nm > 1 nm voxels
    assert util.detector_kernel_kind('auto', 10., 0.248, [1., 1., 1.], (64, 64)) == 'TF'
    with pytest.raises(ValueError):
        util.detector_kernel_kind('exact', 1e3, 0.248, [1., 1., 1.], (64, 64))
    g = np.load(os.path.join(golden_dir, 'g8_kernel_ir_upsample.npz'))
    t = util.device_transfer_function(1000., 0.248, [1., 1., 1.], 32, 32, kernel='IR', dtype=np.complex128)
    np.testing.assert_allclose(t, np.fft.ifftshift(g['Hir_32_32_1000']) / 1024., rtol=0, atol=1e-12 * np.abs(g['Hir_32_32_1000']).max())
    dc = util.transfer_function_dc(1000., 0.248, [1., 1., 1.], 32, 32, kernel='IR')
    assert abs(complex(*dc) - np.fft.ifftshift(g['Hir_32_32_1000'])[0, 0]) <= 1e-12 * abs(complex(*dc))
    with pytest.raises(ValueError):
        util.device_transfer_function(1000., 0.248, [1., 1., 1.], 32, 32, kernel='IR', field_shape=(64, 64))
    # oracle: the IR branch is the same propagation with the other multiplier
    rng = np.random.default_rng(0)
    d = rng.uniform(0, 1e-5, size=(1, 16, 16, 3))
    one, zero = np.ones((16, 16)), np.zeros((16, 16))
    none, _ = orc.multislice_propagate_batch_numpy(d, 0.1 * d, one, zero, 5000., 1e-7, None, d.shape)
    ir, _ = orc.multislice_propagate_batch_numpy(d, 0.1 * d, one, zero, 5000., 1e-7, 1e-4, d.shape, detector_kernel='IR')
    want = orc._propagate(none, orc.get_kernel_ir(1000., 0.248, np.array([1., 1., 1.]), (16, 16, 3)))
    np.testing.assert_allclose(ir, want, rtol=0, atol=1e-13)


def test_summary_columns_and_build_id(tmp_path):
    """summary.txt keeps the reference's '{:<20}{}' layout (cnn_propagator/misc.py:61-77) and still separates a name that fills the
    column from its value; _lib.build_id names what a measurement belongs to."""
    from beyond_dof_amd.misc import create_summary
    from beyond_dof_amd import _lib
    create_summary(str(tmp_path), {'obj_size': (8, 8, 8), 'adjoint_precision': 'first-step', 'adjoint_precision_effective': 'float32'}, preset='ptycho')
    rows = dict(line.split(None, 1) for line in open(str(tmp_path / 'summary.txt')).read().splitlines() if len(line.split(None, 1)) == 2)
    assert rows['obj_size'].strip() == '(8, 8, 8)' and rows['adjoint_precision'].strip() == 'first-step'
    assert rows['adjoint_precision_effective'].strip() == 'float32'
    assert open(str(tmp_path / 'summary.txt')).read().splitlines()[0] == '{:<20}{}'.format('obj_size', '(8, 8, 8)')
    b = _lib.build_id()
    assert set(b) == {'source_sha256', 'flags', 'lib_sha256'} and len(b['source_sha256']) == 16
