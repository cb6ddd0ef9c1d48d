"""The TF twin's bilinear rotation (tf.contrib.image.rotate, tensorflow_recon/fullfield.py:96) restated in the oracle: TF is
not installed, so the restatement is checked against an independent bilinear sampler (scipy.ndimage.map_coordinates) on the
published coordinate map, and its adjoint by the dot-product identity.  Parity unpinned by execution of the reference."""
import numpy as np
from scipy.ndimage import map_coordinates

from oracle import bdof_oracle as orc


def test_rotate_bilinear_matches_an_independent_sampler():
    rng = np.random.default_rng(0)
    obj = rng.normal(size=(3, 9, 12, 2))
    for theta in (0.0, 0.3, -1.1, np.pi / 2, 2.5):
        out = orc.rotate_bilinear(obj, theta)
        c, s, xo, yo = orc.rotate_bilinear_params(theta, 9, 12)
        hh, ww = np.mgrid[:9, :12].astype(float)
        coords = np.stack([s * ww + c * hh + yo, c * ww - s * hh + xo])
        for y in range(3):
            for ch in range(2):
                # pad by one so that taps outside the image read 0 exactly as tf's read_with_fill_value does
                img = np.pad(obj[y, :, :, ch], 1)
                ref = map_coordinates(img, coords + 1, order=1, mode='constant', cval=0.0)
                assert np.abs(out[y, :, :, ch] - ref).max() < 1e-12
    assert np.abs(orc.rotate_bilinear(obj, 0.0) - obj).max() < 1e-12          # identity at angle 0


def test_rotate_bilinear_adjoint_dot_product():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 8, 8, 2))
    y = rng.normal(size=(2, 8, 8, 2))
    for theta in (0.4, -2.0):
        lhs = np.sum(orc.rotate_bilinear(x, theta) * y)
        rhs = np.sum(x * orc.rotate_bilinear_adjoint(y, theta))
        assert abs(lhs - rhs) < 1e-12 * max(1, abs(lhs))
