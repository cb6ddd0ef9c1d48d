"""Pins the oracle's hand-derived gradient, which stands in for
autograd.grad(calculate_loss) (cnn_propagator/fullfield.py:329): finite differences of the
golden-pinned forward, torch autograd (CPU, complex128), and the <Ax,y> = <x,A^H y> identity
of the rotation gather."""
import numpy as np
import pytest
import torch

from oracle import bdof_oracle as orc


def _setup(B=2, Y=8, X=10, S=5, seed=0):
    rng = np.random.default_rng(seed)
    delta = rng.uniform(0, 2e-3, size=(B, Y, X, S))
    beta = rng.uniform(0, 2e-4, size=(B, Y, X, S))
    pr = 1 + 0.1 * rng.normal(size=(Y, X))
    pi = 0.1 * rng.normal(size=(Y, X))
    meas = np.abs(1 + 0.05 * rng.normal(size=(B, Y, X)))
    return delta, beta, pr, pi, meas


def _loss(delta, beta, pr, pi, meas, fp, variant):
    w, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, free_prop_cm=fp,
                                                obj_batch_shape=delta.shape, variant=variant)
    return np.mean((np.abs(w) - meas) ** 2)


@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
@pytest.mark.parametrize('variant', ['numpy_skip_last', 'tf_all'])
def test_gradient_finite_difference(fp, variant):
    delta, beta, pr, pi, meas = _setup()
    if fp == 'inf':
        meas = meas * np.sqrt(meas.shape[1] * meas.shape[2])
    loss, gd, gb = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, fp, variant)
    assert abs(loss - _loss(delta, beta, pr, pi, meas, fp, variant)) < 1e-14 * max(1, abs(loss))
    rng = np.random.default_rng(1)
    for arr, g, which in [(delta, gd, 0), (beta, gb, 1)]:
        for _ in range(6):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            eps = 1e-6
            a_p, a_m = arr.copy(), arr.copy()
            a_p[idx] += eps
            a_m[idx] -= eps
            args_p = (a_p, beta) if which == 0 else (delta, a_p)
            args_m = (a_m, beta) if which == 0 else (delta, a_m)
            fd = (_loss(*args_p, pr, pi, meas, fp, variant) - _loss(*args_m, pr, pi, meas, fp, variant)) / (2 * eps)
            assert abs(fd - g[idx]) <= 1e-6 * max(abs(g).max(), 1e-30), (idx, fd, g[idx])


@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
def test_gradient_vs_torch_autograd(fp):
    delta, beta, pr, pi, meas = _setup(seed=3)
    B, Y, X, S = delta.shape
    loss, gd, gb, g0 = orc.multislice_loss_and_grad(delta, beta, pr, pi, 5000., 1e-7, meas, fp, return_probe_grad=True)
    voxel = np.array([1e-7] * 3) * 1e7
    lmbda = 1240. / 5000.
    h = torch.from_numpy(np.fft.ifftshift(orc.get_kernel(voxel[-1], lmbda, voxel, (Y, X, S))))
    k = 2. * orc.PI * voxel[-1] / lmbda
    td = torch.tensor(delta, requires_grad=True)
    tb = torch.tensor(beta, requires_grad=True)
    # the reference rounds the probe to complex64 (np_funcs.py:20-21)
    p64 = (pr + 1j * pi).astype(np.complex64).astype(np.complex128)
    tpr = torch.tensor(p64.real, requires_grad=True)          # the probe as a variable too (tensorflow_recon/fullfield.py:322-323)
    tpi = torch.tensor(p64.imag, requires_grad=True)
    psi = torch.complex(tpr, tpi).expand(B, Y, X)
    for i in range(S):
        psi = psi * torch.exp(1j * k * td[..., i]) * torch.exp(-k * tb[..., i])
        if i < S - 1:
            psi = torch.fft.ifft2(torch.fft.fft2(psi) * h)
    if fp == 'inf':
        psi = torch.fft.fftshift(torch.fft.fft2(psi), dim=(1, 2))
    elif fp is not None:
        hd = torch.from_numpy(np.fft.ifftshift(orc.get_kernel(fp * 1e7, lmbda, voxel, (Y, X, S))))
        psi = torch.fft.ifft2(torch.fft.fft2(psi) * hd)
    tl = torch.mean((torch.abs(psi) - torch.tensor(meas)) ** 2)
    tl.backward()
    assert abs(tl.item() - loss) < 1e-13 * max(1.0, abs(loss))
    np.testing.assert_allclose(gd, td.grad.numpy(), rtol=0, atol=1e-12 * np.abs(gd).max())
    np.testing.assert_allclose(gb, tb.grad.numpy(), rtol=0, atol=1e-12 * np.abs(gb).max())
    # G(psi_0) summed over the batch = dL/d(probe_real) + i dL/d(probe_imag)
    gp = g0.sum(axis=0)
    np.testing.assert_allclose(gp.real, tpr.grad.numpy(), rtol=0, atol=1e-12 * np.abs(gp).max())
    np.testing.assert_allclose(gp.imag, tpi.grad.numpy(), rtol=0, atol=1e-12 * np.abs(gp).max())


def test_rotation_adjoint_dot_product():
    coords = orc.rotation_lookup([5, 9, 9], 6)
    rng = np.random.default_rng(2)
    x = rng.normal(size=(5, 9, 9, 2))
    y = rng.normal(size=(5, 9, 9, 2))
    for c in coords:
        lhs = np.sum(orc.apply_rotation(x, c) * y)
        rhs = np.sum(x * orc.apply_rotation_adjoint(y, c))
        assert abs(lhs - rhs) < 1e-12 * max(1, abs(lhs))


def test_tv_grad_finite_difference():
    rng = np.random.default_rng(5)
    a = rng.normal(size=(4, 5, 6))
    g = orc.total_variation_3d_grad(a)
    for _ in range(10):
        idx = tuple(rng.integers(0, s) for s in a.shape)
        eps = 1e-7
        ap, am = a.copy(), a.copy()
        ap[idx] += eps
        am[idx] -= eps
        fd = (orc.total_variation_3d(ap) - orc.total_variation_3d(am)) / (2 * eps)
        assert abs(fd - g[idx]) < 1e-6


def test_fullfield_and_ptycho_grad_finite_difference():
    rng = np.random.default_rng(7)
    Y, X, Z = 6, 8, 8
    coords = orc.rotation_lookup([Y, X, Z], 5)
    od = rng.uniform(0, 1e-3, size=(Y, X, Z))
    ob = rng.uniform(0, 1e-4, size=(Y, X, Z))
    prj = 1 + 0.05 * rng.normal(size=(2, Y, X))
    args = (coords, [1, 3], prj, np.ones((Y, X)), np.zeros((Y, X)), 5000., 1e-7)
    kw = dict(free_prop_cm=1e-4, alpha_d=1e-3, alpha_b=1e-3, gamma=1e-3)
    loss, gd, gb = orc.fullfield_loss_and_grad(od, ob, *args, **kw)
    for _ in range(5):
        idx = tuple(rng.integers(0, s) for s in od.shape)
        eps = 1e-7
        p, m = od.copy(), od.copy()
        p[idx] += eps
        m[idx] -= eps
        fd = (orc.fullfield_loss_and_grad(p, ob, *args, **kw)[0] - orc.fullfield_loss_and_grad(m, ob, *args, **kw)[0]) / (2 * eps)
        assert abs(fd - gd[idx]) < 2e-6 * max(1.0, np.abs(gd).max())
    # ptycho
    pos_all = np.array([(y, x) for y in (1, 4) for x in (2, 6)])
    psz = (4, 4)
    prr, pii = orc.gaussian_probe(psz, 2, 2, 0.5)
    meas = np.abs(rng.normal(size=(2, 4, 4))) * 4
    pargs = (coords[2], pos_all, pos_all[[0, 3]], meas, prr, pii, psz, 5000., 1e-7)
    loss, gd, gb = orc.ptycho_loss_and_grad(od, ob, *pargs)
    for arr_i, g in [(0, gd), (1, gb)]:
        for _ in range(5):
            idx = tuple(rng.integers(0, s) for s in od.shape)
            eps = 1e-7
            base = [od, ob]
            p, m = base[arr_i].copy(), base[arr_i].copy()
            p[idx] += eps
            m[idx] -= eps
            ap = [p, ob] if arr_i == 0 else [od, p]
            am = [m, ob] if arr_i == 0 else [od, m]
            fd = (orc.ptycho_loss_and_grad(*ap, *pargs)[0] - orc.ptycho_loss_and_grad(*am, *pargs)[0]) / (2 * eps)
            assert abs(fd - g[idx]) < 2e-6 * max(1.0, np.abs(g).max())


@pytest.mark.parametrize('fp', [None, 1e-4, 'inf'])
def test_cnn_gradient_finite_difference(fp):
    """Real-space truncated-kernel propagator (propagation.py) incl. the corner-pixel renormalisation."""
    rng = np.random.default_rng(5)
    B, Y, X, S, ks = 2, 9, 11, 4, 5
    delta = rng.uniform(0, 2e-3, size=(B, Y, X, S))
    beta = rng.uniform(0, 2e-4, size=(B, Y, X, S))
    pr = 1 + 0.1 * rng.normal(size=(Y, X))
    pi = 0.1 * rng.normal(size=(Y, X))
    meas = np.abs(1 + 0.05 * rng.normal(size=(B, Y, X))) * (np.sqrt(Y * X) if fp == 'inf' else 1.0)
    args = (pr, pi, 5000., [1e-7] * 3, meas)

    def loss_of(d, b):
        w = orc.multislice_propagate_cnn(d, b, pr, pi, 5000., [1e-7] * 3, kernel_size=ks, free_prop_cm=fp)
        return np.mean((np.abs(w) - meas) ** 2)

    loss, gd, gb = orc.cnn_loss_and_grad(delta, beta, *args, kernel_size=ks, free_prop_cm=fp)
    assert abs(loss - loss_of(delta, beta)) < 1e-14 * max(1, loss)
    for arr, g, which in [(delta, gd, 0), (beta, gb, 1)]:
        for trial in range(8):
            idx = (0, 0, 0, int(rng.integers(0, S))) if trial == 0 else tuple(rng.integers(0, s) for s in arr.shape)
            eps = 1e-6
            p, m = arr.copy(), arr.copy()
            p[idx] += eps
            m[idx] -= eps
            fd = ((loss_of(p, beta) - loss_of(m, beta)) if which == 0 else (loss_of(delta, p) - loss_of(delta, m))) / (2 * eps)
            assert abs(fd - g[idx]) <= 2e-6 * max(abs(g).max(), 1e-30), (idx, fd, g[idx])


def test_cnn_kernel_is_separable():
    """The cropped real-space Fresnel kernel is an outer product (H is separable): basis of the GPU's two 1-D passes."""
    k2 = orc.conv_kernel_2d(1.0, 0.248, np.array([1., 1., 1.]), np.array([64, 48, 10]), 17)
    c = 8
    sep = np.outer(k2[:, c], k2[c, :]) / k2[c, c]
    assert np.abs(sep - k2).max() <= 1e-13 * np.abs(k2).max()
