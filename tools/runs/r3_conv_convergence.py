import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/examples')
import reconstruct_phantom as ex
for prop in ('conv', 'fft'):
    r = ex.run(128, 60, 100, 2e-8, 1e-3, quiet=True, propagator=prop)
    print(prop, r)
