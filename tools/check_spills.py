#!/usr/bin/env python
"""Scan the gfx950 ISA of libbdof's kernels for register spills that execute under a restricted EXEC mask.

Why: hipcc 7.2 placed the spill of two long-lived accumulators of k_resident<128> in the flow block of an `if (tid < N)` region,
BEFORE `s_or_b64 exec, exec, s[..]` restored the mask.  Waves without a lane in that region (EXEC = 0) never stored the values
and later reloaded, under the full mask, whatever an earlier kernel had left in that scratch slot — a loss that was 5 % off
depending on what had run before on the CU (round 3).  The pattern is mechanical, so it is checked mechanically:

  within one basic block, a `scratch_store ... Folded Spill` that is followed by `s_or_b64 exec, exec, s[...]` (the end of a
  divergent region) with no instruction in between that narrows or saves the mask again

is reported.  A spill in such a position is only safe if its reload sits under the same mask, which the scan cannot prove;
none at all is the invariant the test suite holds the build to (tests/test_cabi_symbols.py::test_no_spill_under_restricted_exec).

usage: python tools/check_spills.py [extra hipcc flags]      (exit code 1 if any hit)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'beyond_dof_amd', 'csrc', 'bdof_capi.hip')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize']


def device_isa(extra=None):
    if extra is None:
        extra = os.environ.get('BDOF_BUILD_FLAGS', '').split()      # the flags __graft_entry__.build() compiles with
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'bdof.s')
        subprocess.check_call([HIPCC] + BASE_FLAGS + list(extra) + ['--cuda-device-only', '-S', '-o', out, SRC],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


LABEL = re.compile(r'^([.\w$]+):')
KERNEL = re.compile(r'^(_Z\w+):\s*; @')


def scan(isa):
    """[(kernel, line number, text)] of spills stored just before an exec restore."""
    hits = []
    kernel = None
    pending = []           # spills seen in the current basic block since the last instruction that touched exec
    for ln, line in enumerate(isa.splitlines(), 1):
        m = KERNEL.match(line)
        if m:
            kernel, pending = m.group(1), []
            continue
        if LABEL.match(line):
            pending = []
            continue
        t = line.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        if t.startswith('scratch_store') and 'Folded Spill' in t:
            pending.append((ln, t))
        elif t.startswith('s_or_b64 exec, exec,'):
            hits += [(kernel, l, s) for l, s in pending]
            pending = []
        elif 'exec' in t.split(';')[0] and (t.startswith('s_and_saveexec') or t.startswith('s_mov_b64 exec') or t.startswith('s_andn2_b64 exec')
                                            or t.startswith('s_xor_b64 exec') or t.startswith('s_and_b64 exec')):
            pending = []
        elif t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('s_endpgm'):
            pending = []
    return hits


def main():
    hits = scan(device_isa(sys.argv[1:] or None))
    for kernel, ln, text in hits:
        dem = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', kernel], stdout=subprocess.PIPE).stdout.decode().strip() or kernel
        print('{}: line {}: {}'.format(dem, ln, text))
    print('{} spill(s) under a restricted exec mask'.format(len(hits)))
    return 1 if hits else 0


if __name__ == '__main__':
    sys.exit(main())
