"""Tiled ("pfft") Fresnel multislice propagation — the feature the reference repository is named after
(README.md:1-11, "Tiling-based Fresnel multislice propagation"; its scripts live on a branch that is not in the checkout,
so it is checked against the whole-field FFT propagator of np_funcs.py:15-65 on the same field: parity unpinned by reference code).

A (FY, FX) wavefield too large for one fused FFT plan (BASELINE cfg4: a 512^2 probe zero-padded into 4096^2, 1024 slices) is
cut into overlapping T x T tiles, T a fused plan size.  The tiles run through libbdof's per-slice kernels as ONE batch
(bdof_forward_range), each seeing its window of the object (the ptychography window mechanism: origins in xoff / yoff).  A
tile's own FFT is periodic, so errors enter at its edges and move inwards by at most lambda dz / (2 dx^2) pixels per slice (the
steepest ray the grid carries); before they have crossed the halo the cores are stitched back into the field and the tiles
re-cut with fresh halos (bdof_tiles_scatter / bdof_tiles_gather).  The tiles apply the FIELD's transfer function
(util.get_kernel_tile), not that of a T-point mesh.

Long-range correction (long_range=True; round 4).  The band-limited whole-field propagator has alternating tails
~ lambda dz n / (2 pi x^2) after n slices that reach across the whole field; a tile sees no source beyond its halo, and at
BASELINE cfg4's depth (1024 slices) that alone puts 2.2e-5 between the tiled and the whole-field exit wave — in float64,
whatever the stitch interval (tools/cfg4_f64_torch.py).  Free space composes (P^n = F^-1 H^n F), so once per stitch range the
missing part is added back:  psi_out = stitch(T psi_in) + D psi_in,  D = (whole-field free-space step over the range) - (the
tiles' own free-space step, stitched) — one whole-field transform pair and one transform pair per tile per RANGE
(bdof_fields_free_step), exact to first order in the object's phase over a range.  With ranges shorter than the band edge's
phase-winding length, 4 dx^2 / (lambda dz) slices (16 at 5 keV / 1 nm — the default interval with the correction on), the
tiling error at cfg4's depth is 9e-7 (64-pixel halo; 2e-6 with 32).

Precision.  With the correction on, the FIELD is complex128 while it propagates: the whole-field free-space step (rocFFT, double)
carries its bulk from range to range, and the tiles — the fused float32 row kernels, with dithered transform constants AND a
dithered transfer function (bdof_set_transfer_f64: a fixed float32 H is the same perturbation in every slice, 1.4e-5 of the
exit wave after 1024 slices all by itself) — add T psi - T_free psi, the object's part.  cfg4 (4096^2, 1024 slices, 512^2 tiles,
64-pixel halo): 5.8e-6 from the float64 whole field in 343 ms (round 3, no correction, fixed H: 2.7e-5 in 218 ms; the
whole-field float32 rocFFT engine: 1.9e-4 in 620 ms).  precision='float64' runs the tiles themselves in float64
(bdof_forward_range_f64: rocFFT double transforms + point-wise kernels, unfused): 9e-7 in 1.59 s — the reference's own precision
(np_funcs.py:20-42, quirk Q2); 'auto' takes it beyond 2048 slices.

Several GPUs (comm with size > 1): the tiles are dealt to the ranks round-robin; every rank keeps the whole field.  At a
stitch a rank writes its own cores into a zeroed field and the fields are summed over the ranks (every pixel lies in
exactly one core, so the sum only fills in); in the adjoint sweep the tapered scatter-add of the tiles and, at the end, the
object gradient are summed the same way.  One all-reduce of FY x FX complex values per stitch and direction.
"""
import numpy as np

from . import _lib, util
from ._lib import DeviceBuffer
from .comm import PseudoComm
from .engine import MultisliceEngine


class TiledPropagator(object):
    def __init__(self, field_shape, n_slice, energy_ev, psize_cm, tile=512, halo='auto', slices_per_exchange=None, safety=0.5,
                 taper=None, variant='numpy_skip_last', device=0, pi=util.PI, with_grad=False, comm=None, long_range='auto',
                 precision='auto', skip_vacuum=True, carrier='auto'):
        """field_shape (FY, FX); tile: fused plan size (64 ... 1024); halo: pixels per side that are recomputed, not kept; its
        outer `taper` pixels (default halo / 2) are ramped to zero so that the tile's periodic boundary has no jump.  halo='auto':
        64 for plain stitching (ranges of ~130 slices at 5 keV / 1 nm); with the long-range correction, whose ranges are the
        winding length (16 slices), what one range needs — the band edge's reach over it, twice over, plus ramp and margin: 40
        pixels with per-tile carriers (48 at 5 keV / 1 nm; cfg4: 7.7e-7), 16 without (24: the float32 round-off of full-amplitude
        sweeps, 6e-6, hides what more would buy).
        slices_per_exchange: slices between two stitches; default safety * (halo - taper) / (lambda dz / (2 dx^2)), or, with
        the long-range correction, the band edge's phase-winding length 4 dx^2 / (lambda dz) if that is shorter.
        long_range: True / False / 'auto' (on for stacks deeper than one default stitch range, forward model and gradient).
        precision: 'float32' / 'float64' / 'auto' (float64 tiles beyond 2048 slices; forward model only).
        skip_vacuum (forward model with the long-range correction, float32 tiles): a tile whose whole window is vacuum over a
        range contributes T psi - T_free psi = 0 to it — the field's own free-space step already carries the wave there — and is
        left out of that range's launches (a 512^2 zone plate in a 4096^2 padded field: 6 tiles of 81).
        carrier (same mode): every tile of a range rides on its own free-space propagation through the range, formed in double
        (bdof_set_range_carrier); the float32 kernels then carry only the scattered part, whose round-off is what is left of
        T psi - T_free psi.  'auto': in the ranges where at most a quarter of the tiles see any object (a double-precision
        transform pair per slice and tile: cheap for a padded field, 4 x the sweep's time for a field full of object)."""
        self.fy, self.fx = int(field_shape[0]), int(field_shape[1])
        self.n_slice, self.tile = int(n_slice), int(tile)
        if isinstance(halo, str):
            if halo != 'auto':
                raise ValueError("halo: a number of pixels or 'auto'")
            carriers = carrier in ('auto', True) and not with_grad and precision in ('auto', 'float32')
            halo = self._auto_halo(energy_ev, psize_cm, safety, slices_per_exchange, long_range, comm, carriers)
        self.halo = int(halo)
        if 2 * self.halo >= self.tile:
            raise ValueError('the halo must leave a core')
        self.core = self.tile - 2 * self.halo
        self.taper = self.halo // 2 if taper is None else int(taper)
        self.variant = variant
        voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
        lmbda_nm = 1240. / energy_ev
        self.spread_px = lmbda_nm * voxel_nm[2] / (2. * voxel_nm[0] ** 2)      # lateral reach of the band-edge ray per slice
        geometric = max(1, int(safety * (self.halo - self.taper) / self.spread_px)) if self.halo > 0 else self.n_slice
        self.with_grad = bool(with_grad)
        if long_range == 'auto':
            long_range = self.halo > 0 and self.n_slice > geometric and (comm is None or comm.size == 1) and \
                (slices_per_exchange is None or slices_per_exchange < self.n_slice)
        self.long_range = bool(long_range)
        if self.long_range and (comm is not None and comm.size > 1):
            raise ValueError('the long-range correction runs on one rank (it transforms the whole field)')
        if precision == 'auto':
            # fused float32 kernels (dithered transform constants and transfer function): 5.8e-6 at 1024 slices, growing like sqrt(S)
            precision = 'float64' if (self.n_slice > 2048 and not self.with_grad and (comm is None or comm.size == 1)
                                      and carrier is False) else 'float32'
        if precision not in ('float32', 'float64'):
            raise ValueError("precision must be 'float32', 'float64' or 'auto'")
        if precision == 'float64' and (self.with_grad or (comm is not None and comm.size > 1)):
            raise ValueError("precision='float64' is implemented for the forward model on one rank")
        self.precision = precision
        # slices in which the band edge's phase pi lambda dz / (2 dx)^2 winds by pi: beyond it the first-order correction degrades
        self.winding = max(1, int(4. * voxel_nm[0] ** 2 / (lmbda_nm * voxel_nm[2])))
        if slices_per_exchange is None:
            slices_per_exchange = min(geometric, self.winding) if self.long_range else geometric
        self.seg = max(1, int(slices_per_exchange))
        # tile origins: cores tile the field, halos reach periodically across its edges
        ox = [i * self.core - self.halo for i in range(-(-self.fx // self.core))]
        oy = [i * self.core - self.halo for i in range(-(-self.fy // self.core))]
        x0 = np.repeat(np.array(ox, dtype=np.int32), len(oy))
        y0 = np.tile(np.array(oy, dtype=np.int32), len(ox))
        self.n_tiles_field = len(x0)
        self.comm = comm if comm is not None else PseudoComm()
        if self.comm.size > self.n_tiles_field:
            raise ValueError('{} ranks for {} tiles'.format(self.comm.size, self.n_tiles_field))
        self.x0 = np.ascontiguousarray(x0[self.comm.rank::self.comm.size])        # this rank's tiles
        self.y0 = np.ascontiguousarray(y0[self.comm.rank::self.comm.size])
        self.n_tiles = len(self.x0)
        # gradient: tape-free range sweeps with a per-range gradient buffer (bdof_adjoint_range); the ctx holds no [B][S] workspace
        if self.with_grad and variant != 'tf_all':
            raise ValueError("the tiled gradient is written for variant='tf_all' (a transfer-function step after every slice)")
        self.eng = MultisliceEngine(self.tile, self.tile, self.n_slice, self.n_tiles, with_grad=self.with_grad, device=device,
                                    engine='streaming', recompute=self.with_grad, no_grot=self.with_grad)
        self.ctx, self.lib, self.h = self.eng.ctx, self.eng.lib, self.eng.h
        self.comm.attach(self.ctx)
        self.eng.set_physics(energy_ev, psize_cm, None, variant=variant, pi=pi, field_shape=(self.fy, self.fx))
        self.eng.set_probe_none()
        self.idx = DeviceBuffer.from_host(self.ctx, np.stack([np.zeros(self.n_tiles, dtype=np.int32), self.x0, self.y0]))
        self.skip_vacuum = bool(skip_vacuum)
        self._carrier_arg = carrier
        self._active = None               # per stitch range: indices of the tiles whose window holds any object (set_object*)
        self._active_bufs = {}
        n = self.n_tiles * self.tile * self.tile
        self.dbl = precision == 'float64'
        ctype = np.complex128 if self.dbl else np.complex64
        cb = 16 if self.dbl else 8
        self.tiles_in = DeviceBuffer(self.ctx, n * cb, ctype, (self.n_tiles, self.tile, self.tile))
        self.tiles_out = None if self.dbl else DeviceBuffer(self.ctx, n * 8, np.complex64, (self.n_tiles, self.tile, self.tile))
        self.field = DeviceBuffer(self.ctx, self.fx * self.fy * cb, ctype, (self.fx, self.fy))
        self._phys = (voxel_nm[2], lmbda_nm, voxel_nm, pi)
        self.k64 = 2. * pi * voxel_nm[2] / lmbda_nm
        self.h64 = DeviceBuffer.from_host(self.ctx, self._free_table(1, (self.tile, self.tile), np.complex128)) if self.dbl else None
        self._pow_tables = {}
        if self.long_range:
            # float32 tiles: the field itself is kept in complex128 while it propagates (the whole-field free-space step in
            # double carries its bulk, the fused float32 kernels add the object's part); float64 tiles: everything is double
            self.tiles_free = DeviceBuffer(self.ctx, n * cb, ctype, (self.n_tiles, self.tile, self.tile))
            self.whole64 = DeviceBuffer(self.ctx, self.fx * self.fy * 16, np.complex128, (self.fx, self.fy))
            self.field64 = self.field if self.dbl else DeviceBuffer(self.ctx, self.fx * self.fy * 16, np.complex128, (self.fx, self.fy))
            # tiles that see vacuum whatever the object: window origins beyond the volume (the fused free-space step)
            self.vac = DeviceBuffer.from_host(self.ctx, np.stack([np.zeros(self.n_tiles, dtype=np.int32),
                                                                  np.full(self.n_tiles, 1 << 28, dtype=np.int32), self.y0]))
            self._conj_tables = {}
        self.carrier = (self.long_range and not self.dbl and not self.with_grad) if self._carrier_arg == 'auto' else bool(self._carrier_arg)
        if self.carrier and not (self.long_range and not self.dbl and not self.with_grad):
            raise ValueError('carrier fields per tile run in the forward model with the long-range correction and float32 tiles')
        self._car64 = self._car_stack = None
        if self.carrier:
            self._h64_tile = DeviceBuffer.from_host(self.ctx, self._free_table(1, (self.tile, self.tile), np.complex128))

    def _free_table(self, power, shape, ctype, fused_layout=False):
        """The `power`-th power of one slice step's transfer function on a (ny, nx) grid of this field, un-shifted, 1 / (NX NY)
        folded in, formed in float64 (H is exp(i phase): the power is taken on the phase, i.e. on the distance).
        [kx][ky] for bdof_fields_free_step / bdof_forward_range_f64; fused_layout: [ky][kx], the row kernels' order."""
        dist_nm, lmbda_nm, voxel_nm, pi = self._phys
        ny, nx = shape
        if tuple(shape) == (self.fy, self.fx):
            h = util.get_kernel(dist_nm * power, lmbda_nm, voxel_nm, (ny, nx), pi=pi)
        else:
            h = util.get_kernel_tile(dist_nm * power, lmbda_nm, voxel_nm, (ny, nx), (self.fy, self.fx), pi=pi)
        h = np.fft.ifftshift(h) / float(nx * ny)
        return np.ascontiguousarray((h if fused_layout else h.T).astype(ctype))

    def _tables(self, power):
        """(tile table, field table) of `power` consecutive free-space steps, on the device; built once per distinct power.
        The field's is complex128 [kx][ky]; the tiles' complex128 [kx][ky] (float64 tiles) or complex64 [ky][kx] (fused kernels)."""
        if power not in self._pow_tables:
            T = (self.tile, self.tile)
            t = self._free_table(power, T, np.complex128) if self.dbl else self._free_table(power, T, np.complex64, fused_layout=True)
            self._pow_tables[power] = (DeviceBuffer.from_host(self.ctx, t),
                                       DeviceBuffer.from_host(self.ctx, self._free_table(power, (self.fy, self.fx), np.complex128)))
        return self._pow_tables[power]

    # ---- object --------------------------------------------------------------------------------
    def set_object_slab(self, delta2d, beta2d):
        """The same (FY, FX) slab in every slice (a thick zone plate): one set of rows, a table that maps every slice to it."""
        rows = np.ascontiguousarray(np.stack([np.asarray(delta2d).T, np.asarray(beta2d).T], axis=-1).astype(np.float32))     # [x][y] pairs
        vol = DeviceBuffer.from_host(self.ctx, rows)
        tab = np.ascontiguousarray(np.tile(np.arange(self.fx, dtype=np.int32), (1, self.n_slice, 1)))             # [1][S][x] -> row x
        self.eng.set_volume(vol, self.fx, self.fy, DeviceBuffer.from_host(self.ctx, tab), self.fx, 1)
        if getattr(self, 'skip_vacuum', False):             # (WholeFieldPropagator borrows this method: it has no tiles)
            act = self._tiles_with_object((np.asarray(delta2d) != 0) | (np.asarray(beta2d) != 0))
            self._active = [act] * len(self.segments())

    def _tiles_with_object(self, occupied):
        """Indices of this rank's tiles whose (periodic) window touches a pixel of the (FY, FX) mask `occupied`."""
        T = self.tile
        ext = np.pad(occupied.astype(np.int64), ((0, T), (0, T)), mode='wrap')
        sat = np.zeros((ext.shape[0] + 1, ext.shape[1] + 1), dtype=np.int64)
        sat[1:, 1:] = ext.cumsum(axis=0).cumsum(axis=1)
        y = np.mod(self.y0, self.fy)
        x = np.mod(self.x0, self.fx)
        n = sat[y + T, x + T] - sat[y, x + T] - sat[y + T, x] + sat[y, x]
        return np.flatnonzero(n > 0).astype(np.int32)

    def set_object(self, delta, beta):
        """(FY, FX, S) object: rows [x][z][y] of pairs and the identity table row(z, x) = x * S + z."""
        rows = util.volume_to_rows(delta, beta)                                                                   # [X][Z][Y][2]
        vol = DeviceBuffer.from_host(self.ctx, rows)
        x = np.arange(self.fx, dtype=np.int32)
        z = np.arange(self.n_slice, dtype=np.int32)
        tab = np.ascontiguousarray((x[None, :] * self.n_slice + z[:, None])[None].astype(np.int32))               # [1][S][X]
        self.eng.set_volume(vol, self.fx * self.n_slice, self.fy, DeviceBuffer.from_host(self.ctx, tab), self.fx, 1)
        if getattr(self, 'skip_vacuum', False):
            d, b = np.asarray(delta), np.asarray(beta)
            self._active = [self._tiles_with_object(np.any(d[:, :, z0:z0 + nz] != 0, axis=2) | np.any(b[:, :, z0:z0 + nz] != 0, axis=2))
                            for z0, nz in self.segments()]

    def _scattered_range(self, f, w, B, a, xo, yo, z0, nz, prop_last, before_scatter=None):
        """stitch(T psi - T_free psi) of one range added into w (prop_last) / the modulation's scattered part added into f (a last
        single slice without a step), the tiles riding on their own free-space propagation in double."""
        lib, h, T = self.lib, self.h, self.tile
        px = B * T * T
        if self._car64 is None or self._car64.nbytes < px * 16:
            self._car64 = DeviceBuffer(self.ctx, px * 16, np.complex128, (B, T, T))
        if self._car_stack is None or self._car_stack.nbytes < px * 8 * nz:
            self._car_stack = DeviceBuffer(self.ctx, px * 8 * max(nz, self.seg), np.complex64, (max(nz, self.seg), B, T, T))
        car, stack = self._car64, self._car_stack
        self.ctx.check(lib.bdof_tiles_gather_f64(h, f.ptr, self.fx, self.fy, car.ptr, B, T, T, xo, yo, self.taper))
        # p_z = F^-1(H^z F p_0), z = 0 .. nz - 1, in double: one forward transform, one batched inverse one (bdof_range_carrier_build)
        self.ctx.check(lib.bdof_range_carrier_build(h, car.ptr, stack.ptr, B, T, T, self._h64_tile.ptr, nz))
        self.ctx.check(lib.bdof_memset(h, self.tiles_in.ptr, 0, px * 8))              # the scattered part entering the range: none
        self.ctx.check(lib.bdof_set_range_carrier(h, stack.ptr, B, z0, nz))
        try:
            self.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, self.tiles_out.ptr, prop_last))
        finally:
            lib.bdof_set_range_carrier(h, None, 0, 0, 0)
        dst = w if prop_last or nz > 1 else f
        if before_scatter is not None:
            before_scatter()
        self.ctx.check(lib.bdof_tiles_scatter_diff64(h, self.tiles_out.ptr, None, dst.ptr, self.fx, self.fy, B, T, T, xo, yo,
                                                     self.halo, self.halo, 1))

    def _range_tiles(self, i_range):
        """(B, a, xo, yo, va, vx, vy) of stitch range i_range: every tile, or only those that see any object (skip_vacuum)."""
        p, v, B = self.idx.ptr, self.vac.ptr, self.n_tiles
        every = (B, p, p + 4 * B, p + 8 * B, v, v + 4 * B, v + 8 * B)
        if not self.skip_vacuum or self._active is None or self.dbl or self.with_grad:
            return every
        act = self._active[i_range]
        if len(act) == B:
            return every
        if len(act) == 0:
            return (0,) * 7
        key = act.tobytes()
        if key not in self._active_bufs:
            n = len(act)
            zeros = np.zeros(n, dtype=np.int32)
            self._active_bufs[key] = (DeviceBuffer.from_host(self.ctx, np.stack([zeros, self.x0[act], self.y0[act]])),
                                      DeviceBuffer.from_host(self.ctx, np.stack([zeros, np.full(n, 1 << 28, dtype=np.int32), self.y0[act]])))
        ib, vb = self._active_bufs[key]
        n = len(act)
        return (n, ib.ptr, ib.ptr + 4 * n, ib.ptr + 8 * n, vb.ptr, vb.ptr + 4 * n, vb.ptr + 8 * n)

    # ---- forward -------------------------------------------------------------------------------
    def _auto_halo(self, energy_ev, psize_cm, safety, slices_per_exchange, long_range, comm, carriers=False):
        voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
        lmbda_nm = 1240. / energy_ev
        spread = lmbda_nm * voxel_nm[2] / (2. * voxel_nm[0] ** 2)
        plain = min(64, self.tile // 4)
        geometric = max(1, int(safety * (plain - plain // 2) / spread))
        if long_range == 'auto':
            long_range = self.n_slice > geometric and (comm is None or comm.size == 1) and \
                (slices_per_exchange is None or slices_per_exchange < self.n_slice)
        if not long_range:
            return plain
        winding = max(1, int(4. * voxel_nm[0] ** 2 / (lmbda_nm * voxel_nm[2])))
        seg = winding if slices_per_exchange is None else int(slices_per_exchange)
        # without per-tile carriers the float32 round-off of the sweeps (6e-6 at cfg4's depth) hides what a wider ramp would buy;
        # with them the tiling error is what is left, and 40 pixels of ramp and margin bring it under 1e-6
        need = 2 * int(np.ceil(spread * seg / safety)) + (40 if carriers else 16)
        return int(min(max(need, 16), plain))

    def segments(self):
        return [(z0, min(self.seg, self.n_slice - z0)) for z0 in range(0, self.n_slice, self.seg)]

    def forward_device(self):
        """Propagate the device field in place through all slices (np_funcs.py:36-43: no step after the last slice unless
        variant == 'tf_all')."""
        lib, h, p = self.lib, self.h, self.idx.ptr
        a, xo, yo = p, p + 4 * self.n_tiles, p + 8 * self.n_tiles
        B, T = self.n_tiles, self.tile
        if not self.long_range:
            gather = lib.bdof_tiles_gather_f64 if self.dbl else lib.bdof_tiles_gather
            for z0, nz in self.segments():
                prop_last = int(z0 + nz < self.n_slice or self.variant == 'tf_all')
                self.ctx.check(gather(h, self.field.ptr, self.fx, self.fy, self.tiles_in.ptr, B, T, T, xo, yo, self.taper))
                if self.dbl:
                    self.ctx.check(lib.bdof_forward_range_f64(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, self.h64.ptr, self.k64, prop_last))
                    self._stitch(self.tiles_in)
                else:
                    self.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, self.tiles_out.ptr, prop_last))
                    self._stitch(self.tiles_out)
            return
        # Long-range correction: psi_out = P_field^n psi_in + stitch(T psi_in - T_free psi_in) per range — the field's own
        # free-space step over the range (one transform pair of the whole field, in double) carries the wave, the tiles add what
        # the object does to it; T_free is the tiles' own free-space step, so what the tiles miss of the field's propagator
        # (its long-range tails) drops out to first order in the object's phase over one range.
        v = self.vac.ptr
        va, vx, vy = v, v + 4 * B, v + 8 * B
        npx = self.fx * self.fy
        if not self.dbl:
            self.ctx.check(lib.bdof_c_convert(h, self.field64.ptr, self.field.ptr, npx, 1))
        f, w = self.field64, self.whole64
        for i_range, (z0, nz) in enumerate(self.segments()):
            prop_last = int(z0 + nz < self.n_slice or self.variant == 'tf_all')
            nprop = nz - 1 + prop_last                       # transfer-function steps of this range
            B, a, xo, yo, va, vx, vy = self._range_tiles(i_range)      # (tiles whose window is vacuum over the range add nothing)
            if B == 0:
                if nprop:
                    _, f_tab = self._tables(nprop)
                    self.ctx.check(lib.bdof_fields_free_step(h, f.ptr, 1, self.fx, self.fy, f_tab.ptr, 0, 1))
                continue
            # per-tile carriers cost one double-precision transform pair per slice and tile: 'auto' takes them where few tiles see
            # the object (a padded field), and sweeps the full wave in float32 where most do (4 x the time for 2.4 x the accuracy)
            if self.carrier and (self._carrier_arg is True or 4 * B <= self.n_tiles):
                if nprop == 0:                               # a last single slice without a step: f += (c - 1) psi on the cores
                    self._scattered_range(f, f, B, a, xo, yo, z0, nz, 0)
                    continue
                # the field's own step (one double transform pair of the whole field) on the auxiliary stream, beside the few
                # tiles' carrier steps and sweeps: the one saturates HBM, the others are small launches
                _, f_tab = self._tables(nprop)
                self.ctx.check(lib.bdof_fields_free_step_aux(h, w.ptr, f.ptr, 1, self.fx, self.fy, f_tab.ptr, 0, 1))
                try:
                    self._scattered_range(f, w, B, a, xo, yo, z0, nz, prop_last, before_scatter=lambda: self.ctx.check(lib.bdof_aux_join(h)))
                finally:
                    lib.bdof_aux_join(h)                     # (a no-op after the join above; an error on the way must not leave it open)
                f, w = w, f
                continue
            if self.dbl:
                self.ctx.check(lib.bdof_tiles_gather_f64(h, f.ptr, self.fx, self.fy, self.tiles_in.ptr, B, T, T, xo, yo, self.taper))
            else:
                self.ctx.check(lib.bdof_tiles_gather_mixed(h, f.ptr, self.fx, self.fy, self.tiles_in.ptr, B, T, T, xo, yo, self.taper))
            if nprop == 0:                                   # a last single slice without a step: nothing to correct
                if self.dbl:
                    self.ctx.check(lib.bdof_forward_range_f64(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, self.h64.ptr, self.k64, 0))
                    self.ctx.check(lib.bdof_tiles_scatter_f64(h, self.tiles_in.ptr, f.ptr, self.fx, self.fy, B, T, T, xo, yo, self.halo, self.halo))
                else:
                    self.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, self.tiles_out.ptr, 0))
                    self.ctx.check(lib.bdof_tiles_scatter_diff64(h, self.tiles_out.ptr, None, f.ptr, self.fx, self.fy, B, T, T, xo, yo,
                                                                 self.halo, self.halo, 0))
                continue
            t_tab, f_tab = self._tables(nprop)
            self.ctx.check(lib.bdof_memcpy_d2d(h, w.ptr, f.ptr, npx * 16))
            self.ctx.check(lib.bdof_fields_free_step(h, w.ptr, 1, self.fx, self.fy, f_tab.ptr, 0, 1))
            if self.dbl:
                self.ctx.check(lib.bdof_memcpy_d2d(h, self.tiles_free.ptr, self.tiles_in.ptr, self.tiles_in.nbytes))
                self.ctx.check(lib.bdof_fields_free_step(h, self.tiles_free.ptr, B, T, T, t_tab.ptr, 0, 1))
                self.ctx.check(lib.bdof_forward_range_f64(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, self.h64.ptr, self.k64, prop_last))
                self.ctx.check(lib.bdof_caxpy(h, self.tiles_in.ptr, self.tiles_free.ptr, -1.0, B * T * T, 1))
                self.ctx.check(lib.bdof_tiles_scatter_f64(h, self.tiles_in.ptr, f.ptr, self.fx, self.fy, B, T, T, xo, yo, self.halo, self.halo))
                self.ctx.check(lib.bdof_caxpy(h, w.ptr, f.ptr, 1.0, npx, 1))        # (f holds the difference on every core = everywhere)
            else:
                # T_free through the SAME fused kernels (one vacuum slice stepped with H^n): both terms of the difference carry
                # the same dithered constants, and no rocFFT float32 drift enters
                self.ctx.check(lib.bdof_forward_range_h(h, B, va, vx, vy, z0, 1, self.tiles_in.ptr, self.tiles_free.ptr, 1, t_tab.ptr))
                self.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, self.tiles_out.ptr, prop_last))
                self.ctx.check(lib.bdof_tiles_scatter_diff64(h, self.tiles_out.ptr, self.tiles_free.ptr, w.ptr, self.fx, self.fy, B, T, T, xo, yo,
                                                             self.halo, self.halo, 1))
            f, w = w, f
        if f is not self.field64:                            # an odd number of swaps: the result sits in the other buffer
            self.ctx.check(lib.bdof_memcpy_d2d(h, self.field64.ptr, f.ptr, npx * 16))
        if not self.dbl:
            self.ctx.check(lib.bdof_c_convert(h, self.field.ptr, self.field64.ptr, npx, 0))

    def _sum_over_ranks(self, buf):
        if self.comm.size > 1:
            self.comm.allreduce_sum_device(self.ctx, buf)

    def _stitch(self, tiles):
        """Cores of this rank's tiles -> field; with several ranks the other ranks' cores arrive by summation."""
        p = self.idx.ptr
        if self.comm.size > 1:
            self.ctx.check(self.lib.bdof_memset(self.h, self.field.ptr, 0, self.field.nbytes))
        scatter = self.lib.bdof_tiles_scatter_f64 if self.dbl else self.lib.bdof_tiles_scatter
        self.ctx.check(scatter(self.h, tiles.ptr, self.field.ptr, self.fx, self.fy, self.n_tiles, self.tile, self.tile,
                               p + 4 * self.n_tiles, p + 8 * self.n_tiles, self.halo, self.halo))
        self._sum_over_ranks(self.field)

    def forward(self, probe_real, probe_imag):
        """Exit wave (FY, FX) of the probe (FY, FX) through the object: complex64, or complex128 with precision='float64'.
        The probe is rounded to complex64 first, as np_funcs.py:20-21 does."""
        probe = (np.asarray(probe_real) + 1j * np.asarray(probe_imag)) * np.ones((self.fy, self.fx))
        self.field.upload(np.ascontiguousarray(probe.T.astype(np.complex64).astype(self.field.dtype)))
        self.forward_device()
        self.ctx.sync()
        return np.ascontiguousarray(self.field.download().T)

    # ---- loss + gradient -----------------------------------------------------------------------
    def loss_and_grad_device(self, meas_dev):
        """The device part of loss_and_grad: self.field holds the probe [x][y] on entry and G(probe) on exit; meas_dev: device
        float [x][y]; returns (loss, device buffer of the volume gradient in the object's row layout)."""
        if not self.with_grad:
            raise RuntimeError('TiledPropagator(with_grad=True) needed')
        if self.long_range:
            return self._loss_and_grad_long_range(meas_dev)
        import ctypes
        lib, h, p = self.lib, self.h, self.idx.ptr
        a, xo, yo = p, p + 4 * self.n_tiles, p + 8 * self.n_tiles
        T, B = self.tile, self.n_tiles
        segs = self.segments()
        if getattr(self, '_ends', None) is None or len(self._ends) != len(segs):
            self._ends = [DeviceBuffer(self.ctx, B * T * T * 8, np.complex64, (B, T, T)) for _ in segs]
            self._grot = DeviceBuffer(self.ctx, B * max(nz for _, nz in segs) * T * T * 8, np.float32)
            self._gvol = DeviceBuffer.zeros(self.ctx, self.eng._keep['obj'].shape, np.float32)
        for (z0, nz), end in zip(segs, self._ends):
            self.ctx.check(lib.bdof_tiles_gather(h, self.field.ptr, self.fx, self.fy, self.tiles_in.ptr, B, T, T, xo, yo, self.taper))
            self.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, end.ptr, 1))
            self._stitch(end)
        self.ctx.check(lib.bdof_field_loss_seed(h, self.field.ptr, _lib._ptr(meas_dev), self.fx, self.fy))
        gvol = self._gvol
        self.ctx.check(lib.bdof_memset(h, gvol.ptr, 0, gvol.nbytes))
        for (z0, nz), end in reversed(list(zip(segs, self._ends))):
            self.ctx.check(lib.bdof_tiles_scatter_adjoint(h, self.field.ptr, self.fx, self.fy, self.tiles_in.ptr, B, T, T, xo, yo, self.halo, self.halo))
            self.ctx.check(lib.bdof_adjoint_range(h, B, a, xo, yo, z0, nz, end.ptr, self.tiles_in.ptr, self.tiles_out.ptr, self._grot.ptr))
            self.ctx.check(lib.bdof_tiles_grad_add(h, self._grot.ptr, gvol.ptr, B, T, T, xo, yo, z0, nz))
            self.ctx.check(lib.bdof_tiles_gather_adjoint(h, self.tiles_out.ptr, self.field.ptr, self.fx, self.fy, B, T, T, xo, yo, self.taper))
            self._sum_over_ranks(self.field)
        self._sum_over_ranks(gvol)
        loss = ctypes.c_double(0)
        self.ctx.check(lib.bdof_get_loss(h, ctypes.byref(loss)))
        return loss.value, gvol

    def _conj_table(self, power):
        """conj(H^power) for the tiles, in the fused kernels' layout: the adjoint of their free-space step over a range."""
        if power not in self._conj_tables:
            self._conj_tables[power] = DeviceBuffer.from_host(
                self.ctx, np.conj(self._free_table(power, (self.tile, self.tile), np.complex128, fused_layout=True)).astype(np.complex64))
        return self._conj_tables[power]

    def _loss_and_grad_long_range(self, meas_dev):
        """loss_and_grad_device through the CORRECTED tiled model  psi_out = W psi_in + Stitch(T Cut psi_in - F Cut psi_in)  per range
        (W: the field's free-space step over the range, in double; Cut: tapered periodic cut-out; T: the tiles' multislice
        sweep; F: their free-space step; Stitch: cores back).  Adjoint, last range first:
            G_in = W^H G_out + Cut^H ( T^H - F^H ) Stitch^H G_out ,
        T^H by bdof_adjoint_range (which also leaves the object-gradient rows), F^H the fused free-space step with conj(H^n),
        W^H bdof_fields_free_step(conj_h = 1); the field-level adjoint stays in complex128 like the field."""
        import ctypes
        lib, h, p = self.lib, self.h, self.idx.ptr
        a, xo, yo = p, p + 4 * self.n_tiles, p + 8 * self.n_tiles
        v = self.vac.ptr
        va, vx, vy = v, v + 4 * self.n_tiles, v + 8 * self.n_tiles
        T, B, npx = self.tile, self.n_tiles, self.fx * self.fy
        segs = self.segments()
        if getattr(self, '_ends', None) is None or len(self._ends) != len(segs):
            self._ends = [DeviceBuffer(self.ctx, B * T * T * 8, np.complex64, (B, T, T)) for _ in segs]
            self._grot = DeviceBuffer(self.ctx, B * max(nz for _, nz in segs) * T * T * 8, np.float32)
            self._gvol = DeviceBuffer.zeros(self.ctx, self.eng._keep['obj'].shape, np.float32)
        f, w = self.field64, self.whole64
        self.ctx.check(lib.bdof_c_convert(h, f.ptr, self.field.ptr, npx, 1))
        for (z0, nz), end in zip(segs, self._ends):
            t_tab, f_tab = self._tables(nz)                   # tf_all: a step after every slice
            self.ctx.check(lib.bdof_tiles_gather_mixed(h, f.ptr, self.fx, self.fy, self.tiles_in.ptr, B, T, T, xo, yo, self.taper))
            self.ctx.check(lib.bdof_memcpy_d2d(h, w.ptr, f.ptr, npx * 16))
            self.ctx.check(lib.bdof_fields_free_step(h, w.ptr, 1, self.fx, self.fy, f_tab.ptr, 0, 1))
            self.ctx.check(lib.bdof_forward_range_h(h, B, va, vx, vy, z0, 1, self.tiles_in.ptr, self.tiles_free.ptr, 1, t_tab.ptr))
            self.ctx.check(lib.bdof_forward_range(h, B, a, xo, yo, z0, nz, self.tiles_in.ptr, end.ptr, 1))
            self.ctx.check(lib.bdof_tiles_scatter_diff64(h, end.ptr, self.tiles_free.ptr, w.ptr, self.fx, self.fy, B, T, T, xo, yo,
                                                         self.halo, self.halo, 1))
            f, w = w, f
        self.ctx.check(lib.bdof_c_convert(h, self.field.ptr, f.ptr, npx, 0))
        self.ctx.check(lib.bdof_field_loss_seed(h, self.field.ptr, _lib._ptr(meas_dev), self.fx, self.fy))
        g, w = f, w                                            # the field-level adjoint takes over the two float64 buffers
        self.ctx.check(lib.bdof_c_convert(h, g.ptr, self.field.ptr, npx, 1))
        gvol = self._gvol
        self.ctx.check(lib.bdof_memset(h, gvol.ptr, 0, gvol.nbytes))
        for (z0, nz), end in reversed(list(zip(segs, self._ends))):
            _, f_tab = self._tables(nz)
            self.ctx.check(lib.bdof_tiles_scatter_adjoint_mixed(h, g.ptr, self.fx, self.fy, self.tiles_in.ptr, B, T, T, xo, yo, self.halo, self.halo))
            # F^H Stitch^H G: vacuum, conj(H^n)
            self.ctx.check(lib.bdof_forward_range_h(h, B, va, vx, vy, z0, 1, self.tiles_in.ptr, self.tiles_free.ptr, 1, self._conj_table(nz).ptr))
            # T^H Stitch^H G, and the object gradient of the range
            self.ctx.check(lib.bdof_adjoint_range(h, B, a, xo, yo, z0, nz, end.ptr, self.tiles_in.ptr, self.tiles_out.ptr, self._grot.ptr))
            self.ctx.check(lib.bdof_tiles_grad_add(h, self._grot.ptr, gvol.ptr, B, T, T, xo, yo, z0, nz))
            # W^H G in place, then += Cut^H (T^H - F^H) ...
            self.ctx.check(lib.bdof_fields_free_step(h, g.ptr, 1, self.fx, self.fy, f_tab.ptr, 1, 1))
            self.ctx.check(lib.bdof_tiles_gather_adjoint_diff64(h, self.tiles_out.ptr, self.tiles_free.ptr, g.ptr, self.fx, self.fy, B, T, T, xo, yo,
                                                                self.taper, 1))
        self.ctx.check(lib.bdof_c_convert(h, self.field.ptr, g.ptr, npx, 0))            # G(probe), as the plain path leaves it
        loss = ctypes.c_double(0)
        self.ctx.check(lib.bdof_get_loss(h, ctypes.byref(loss)))
        return loss.value, gvol

    def loss_and_grad(self, probe_real, probe_imag, meas_abs):
        """loss = mean((|exit wave| - meas_abs)^2) over the field (fullfield.py:106, no detector step) and its gradient w.r.t.
        the object, through the tiled forward model — the exact adjoint of what forward() computes, range by range, last
        range first: adjoint of the stitch (cores) -> bdof_adjoint_range (the forward wave is marched back from the range's
        end state, no tape) -> gradient rows added into the volume gradient -> adjoint of the tapered cut.  Returns
        (loss, g_delta, g_beta) in the shape the object was given in, and leaves G(probe) in self.field."""
        probe = (np.asarray(probe_real) + 1j * np.asarray(probe_imag)) * np.ones((self.fy, self.fx))
        self.field.upload(np.ascontiguousarray(probe.T.astype(np.complex64)))
        meas = DeviceBuffer.from_host(self.ctx, np.ascontiguousarray(np.asarray(meas_abs, dtype=np.float32).T))
        loss, gvol = self.loss_and_grad_device(meas)
        self.ctx.sync()
        g = gvol.download()
        if g.ndim == 3:                                            # slab rows [x][y][2]
            return loss, np.ascontiguousarray(g[..., 0].T), np.ascontiguousarray(g[..., 1].T)
        gd, gb = util.rows_to_volume(g)
        return loss, gd, gb


class WholeFieldPropagator(object):
    """np_funcs.py:36-43 on the WHOLE (FY, FX) field in float64 on the device: c = exp(i k delta) exp(-k beta) from the object's
    rows, rocFFT double-precision transforms of the field, the field's own transfer function (bdof_forward_range_f64 with one
    "tile" that is the field).  What the tiled propagator is measured against at depths the host cannot reach in a test
    (1024 slices of 4096^2 in complex128 take 393 s of host FFTs, 3 s here); itself checked against the host's float64
    run at 96 slices (tests/test_gpu_tiling.py)."""

    def __init__(self, field_shape, n_slice, energy_ev, psize_cm, variant='numpy_skip_last', device=0, pi=util.PI):
        self.fy, self.fx, self.n_slice, self.variant = int(field_shape[0]), int(field_shape[1]), int(n_slice), variant
        voxel_nm = np.array([psize_cm] * 3) * 1.e7 if np.isscalar(psize_cm) else np.array(psize_cm) * 1.e7
        lmbda_nm = 1240. / energy_ev
        self.eng = MultisliceEngine(self.fy, self.fx, self.n_slice, 1, with_grad=False, device=device, engine='generic')
        self.ctx, self.lib, self.h = self.eng.ctx, self.eng.lib, self.eng.h
        self.k64 = 2. * pi * voxel_nm[2] / lmbda_nm
        hk = np.fft.ifftshift(util.get_kernel(voxel_nm[2], lmbda_nm, voxel_nm, (self.fy, self.fx), pi=pi)) / float(self.fx * self.fy)
        self.h64 = DeviceBuffer.from_host(self.ctx, np.ascontiguousarray(hk.T.astype(np.complex128)))      # [kx][ky]
        self.field = DeviceBuffer(self.ctx, self.fx * self.fy * 16, np.complex128, (self.fx, self.fy))
        self.idx = DeviceBuffer.from_host(self.ctx, np.zeros(3, dtype=np.int32))

    set_object_slab = TiledPropagator.set_object_slab
    set_object = TiledPropagator.set_object

    def forward(self, probe_real, probe_imag):
        probe = (np.asarray(probe_real) + 1j * np.asarray(probe_imag)) * np.ones((self.fy, self.fx))
        self.field.upload(np.ascontiguousarray(probe.T.astype(np.complex64).astype(np.complex128)))
        p = self.idx.ptr
        self.ctx.check(self.lib.bdof_forward_range_f64(self.h, 1, p, p + 4, p + 8, 0, self.n_slice, self.field.ptr, self.h64.ptr, self.k64,
                                                       int(self.variant == 'tf_all')))
        self.ctx.sync()
        return np.ascontiguousarray(self.field.download().T)
