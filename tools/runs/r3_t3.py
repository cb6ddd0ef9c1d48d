import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    from oracle import bdof_oracle as orc
    from beyond_dof_amd import engine
    sizes = [int(v) for v in sys.argv[1].split(',')]
    keep = []
    for n in sizes:
        B, S, fp = 3, 5, None
        rng = np.random.default_rng(n)
        delta = rng.uniform(0, 2e-5, size=(B, n, n, S)); beta = 0.1 * delta
        pr, pi = 1 + 0.1 * rng.normal(size=(n, n)), 0.1 * rng.normal(size=(n, n))
        eng = engine.MultisliceEngine(n, n, S, B, with_grad=True, engine='resident')
        eng.set_physics(5000., 1e-7, fp); eng.set_probe(pr, pi); eng.set_object_batch(delta, beta)
        wave = eng.forward(B)
        ref, _ = orc.multislice_propagate_batch_numpy(delta, beta, pr, pi, 5000., 1e-7, fp, delta.shape, return_probe_array=False)
        meas = np.abs(ref) * (1 + 0.05 * rng.normal(size=ref.shape))
        l1 = eng.loss_grad(B, meas)
        l2 = eng.loss_grad(B, meas)
        rl = np.mean((np.abs(ref) - meas) ** 2)
        print(n, 'wave err', np.linalg.norm(wave - ref) / np.linalg.norm(ref), 'loss', l1, l2, 'oracle', rl, 'rel', abs(l1 - rl) / rl, flush=True)
        if '--keep' in sys.argv:
            keep.append(eng)
    sys.exit(0)
for seq in ('128', '32,128', '36,128', '48,128', '64,128', '72,128', '80,128', '96,128', '32,36,48,64,72,80,96,128'):
    for extra in ([], ['--keep']):
        r = subprocess.run([sys.executable, __file__, seq] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        print('---', seq, extra, 'rc', r.returncode)
        print(r.stdout.decode()[-1500:])
