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


VMEM = ('global_', 'scratch_', 'buffer_', 'flat_')


def scan_counted_waits(isa, kernel_substr='k_conv2'):
    """The counted wait of k_conv2 (csrc/bdof_conv2.h): `s_waitcnt vmcnt(N)` at the end of a tile, issued by asm, stands for "all
    vector-memory operations but this tile's N stores have retired" — true only if the N youngest operations at that point ARE
    stores.  Returns [(kernel, line, why)] for every asm `s_waitcnt vmcnt(N > 0)` of such a kernel that is not directly preceded,
    within its basic block, by at least N vector-memory instructions that are all global stores; and an entry for every such
    kernel that has no counted wait at all.  hipcc is free to merge, split or move memory operations: this is the check that it
    has not."""
    hits = []
    lines = isa.splitlines()
    kernel, start = None, 0
    found = {}
    for ln, line in enumerate(lines):
        m = KERNEL.match(line)
        if m:
            kernel, start = m.group(1), ln
            if kernel_substr in kernel:
                found.setdefault(kernel, 0)
            continue
        if kernel is None or kernel_substr not in kernel:
            continue
        t = line.strip()
        m = re.match(r's_waitcnt vmcnt\((\d+)\)$', t)
        if not m or int(m.group(1)) == 0 or ln == 0 or 'ASMSTART' not in lines[ln - 1]:
            continue
        n = int(m.group(1))
        found[kernel] += 1
        seen = 0
        why = None
        for back in range(ln - 1, start, -1):
            b = lines[back].strip()
            if LABEL.match(lines[back]):
                why = 'only {} vector-memory instructions between the block\'s start and vmcnt({})'.format(seen, n)
                break
            if b.startswith(VMEM):
                if not b.startswith('global_store'):
                    why = 'a {} among the {} youngest vector-memory instructions before vmcnt({})'.format(b.split()[0], n, n)
                    break
                seen += 1
                if seen == n:
                    break
        if why:
            hits.append((kernel, ln + 1, why))
    hits += [(k, 0, 'no counted wait found') for k, c in found.items() if c == 0]
    return hits, len(found)


def main():
    isa = device_isa(sys.argv[1:] or None)
    bad, n = scan_counted_waits(isa)
    for kernel, ln, why in bad:
        print('{}: line {}: {}'.format(kernel, ln, why))
    print('{} counted wait(s) of {} k_conv2 instance(s) not backed by their stores'.format(len(bad), n))
    hits = scan(isa)
    for kernel, ln, text in hits:
        dem = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', kernel], stdout=subprocess.PIPE).stdout.decode().strip() or kernel
        print('{}: line {}: {}'.format(dem, ln, text))
    print('{} spill(s) under a restricted exec mask'.format(len(hits)))
    return 1 if hits or bad else 0


if __name__ == '__main__':
    sys.exit(main())
