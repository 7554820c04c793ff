#!/usr/bin/env python3
"""Dev tool: FLAT (generic-pointer) memory instructions per device function, and how many of them sit inside loops, from
`hipcc -S --offload-device-only` output.  A FLAT access counts on lgkmcnt as well as vmcnt and returns out of order with LDS
operations, so while one is in flight every LDS wait has to be lgkmcnt(0); global_* / ds_* do not.

usage: python3 tools/isa_flat_check.py <asm file> [substring of the mangled names to list, default: fig_hot]"""
import re, sys

def main():
    src = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else "fig_hot"
    names = [n for n in re.findall(r'\n(_Z\w+):', src) if pat in n]
    print('# function | instructions | flat | global | ds | flat inside loops | flat inside innermost loops')
    for nm in names:
        body = re.split(r'\n(?=%s:)' % re.escape(nm), src)[1].split('.Lfunc_end')[0]
        lines = body.split('\n')
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
        ops = [(i, l.strip().split()[0]) for i, l in enumerate(lines) if l.startswith('\t') and not l.startswith('\t.') and l.strip()]
        flat = [i for i, o in ops if o.startswith('flat_')]
        inloop = sum(1 for i in flat if any(a <= i <= b for a, b in loops))
        inner = [(a, b) for a, b in loops if not any(a <= c and d <= b and (c, d) != (a, b) for c, d in loops)]
        ininner = sum(1 for i in flat if any(a <= i <= b for a, b in inner))
        print('%-62s %6d %5d %6d %5d %5d %5d' % (nm[:62], len(ops), len(flat), sum(o.startswith('global_') for _, o in ops), sum(o.startswith('ds_') for _, o in ops), inloop, ininner))

if __name__ == '__main__':
    main()
