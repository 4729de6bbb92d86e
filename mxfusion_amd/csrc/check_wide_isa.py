#!/usr/bin/env python
"""Guard for gemm_f16x2_wide_kernel (gemm_split.hip): its B fragments are fetched with inline-asm global loads whose completion the
compiler does not track, so nothing may touch a destination register between the load and the s_waitcnt that retires it -- in particular
no compiler-inserted register copy (seen once: the copies of a control-flow merge were placed in front of the wait).

usage: check_wide_isa.py file.s     (the gfx950 assembly of gemm_split.hip: hipcc -S --cuda-device-only ...)
Linear scan in layout order of the kernel body: every VMEM request enters a FIFO (vmcnt retires in order), `s_waitcnt vmcnt(N)` retires
all but the youngest N, and any other instruction naming a VGPR inside a still-pending destination range is an error."""
import re
import sys

KERNEL = 'gemm_f16x2_wide_kernel'


def vregs(text):
    out = []
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', text):
        out.append((int(a), int(b)))
    for a in re.findall(r'\bv(\d+)\b', text):
        out.append((int(a), int(a)))
    return out


def kernel_starts(lines):
    """line indices of the entry labels of every instantiation of the kernel template"""
    return [i for i, l in enumerate(lines) if KERNEL in l and not l.startswith(('\t', ' ', '.')) and l.split(';')[0].rstrip().endswith(':')]


def check(lines, start):
    errors, pending, n_loads, n_waits = [], [], 0, 0
    skipping = False
    for ln in lines[start + 1:]:
        s = ln.strip()
        # the fused reverse pass (wide_body<..., FUSE>) is ordinary compiler-tracked code with control flow of its own, between two markers: no
        # inline-asm load may be in flight when it begins, and the linear scan resumes behind it
        if 'mxf_fz_epilogue_begin' in s:
            if any(p for p in pending):
                errors.append('an inline-asm load is still in flight at the start of the fused epilogue')
            skipping, pending = True, []
            continue
        if 'mxf_fz_epilogue_end' in s:
            skipping = False
            continue
        if skipping:
            if s.split(';')[0].strip().startswith('s_endpgm'):
                break
            continue
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        s = s.split(';')[0].strip()
        op = s.split()[0]
        if op == 's_endpgm':
            break
        if op.startswith('global_load_lds') or op.startswith('buffer_load') and ' lds' in s:
            pending.append(None)
            continue
        if op.startswith('global_load') or op.startswith('scratch_load') or op.startswith('buffer_load'):
            regs = vregs(s)
            if not regs:                      # a spill reload into an AGPR / an address held in SGPRs only: no VGPR involved
                pending.append(None)
                continue
            dst, srcs = regs[0], regs[1:]
            for r in srcs:
                if any(p and not (r[1] < p[0] or r[0] > p[1]) for p in pending):
                    errors.append('address of `%s` reads a pending load destination' % s)
            pending.append(dst)
            n_loads += 1
            continue
        if op.startswith('global_store') or op.startswith('scratch_store') or op.startswith('global_atomic') or op.startswith('buffer_store'):
            for r in vregs(s):
                if any(p and not (r[1] < p[0] or r[0] > p[1]) for p in pending):
                    errors.append('`%s` reads a pending load destination' % s)
            pending.append(None)
            continue
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', s)
            if m:
                n = int(m.group(1))
                n_waits += 1
                while len(pending) > n:
                    pending.pop(0)
            continue
        for r in vregs(s):
            if any(p and not (r[1] < p[0] or r[0] > p[1]) for p in pending):
                errors.append('`%s` touches v[%d:%d] while its load is in flight' % (s, r[0], r[1]))
                break
    return errors, n_loads, n_waits


if __name__ == '__main__':
    lines = open(sys.argv[1]).read().splitlines()
    starts = kernel_starts(lines)
    bad = not starts
    if not starts:
        print('%s: no instantiation found in %s' % (KERNEL, sys.argv[1]))
    for st in starts:
        errs, nl, nw = check(lines, st)
        print('%s: %d register loads, %d vmcnt waits, %d violations' % (lines[st].split(':')[0], nl, nw, len(errs)))
        for e in errs[:20]:
            print('  ', e)
        bad = bad or bool(errs) or nl == 0
    sys.exit(1 if bad else 0)
