#!/usr/bin/env python3
"""Dev tool: in the compiled ISA (`hipcc -S --offload-device-only`), every function that enters VGPR-index mode
(`s_set_gpr_idx_on`, the indexed multiplies of fig_engine_shared.h) may touch M0 only through the three instructions of those
blocks: `s_set_gpr_idx_on` (writes M0), `s_and_b32 m0, ..., 0xffff` and `s_lshr_b32 m0, ..., 16`.  Any other read or write of M0
in such a function would be a compiler use of the register the inline asm overwrites (M0 cannot be listed as a
clobber, the backend reserves it).  Exit code 1 when one is found.
usage: python3 tools/isa_m0_check.py <asm file>"""
import re, sys

def main():
    src = open(sys.argv[1]).read()
    bad = 0; checked = 0
    for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', src, re.S):
        name, body = m.group(1), m.group(2)
        if 's_set_gpr_idx_on' not in body:
            continue
        checked += 1
        for ln in body.split('\n'):
            t = ln.strip()
            if not re.search(r'\bm0\b', t) or t.startswith(';') or t.startswith('.'):
                continue
            if re.match(r's_and_b32 m0, s\d+, 0xffff', t) or re.match(r's_lshr_b32 m0, s\d+, 16', t):
                continue
            print(f"{name[:70]}: unexpected M0 use: {t}")
            bad += 1
    print(f"# {checked} functions enter index mode; {bad} M0 accesses outside the indexed-multiply blocks")
    return 1 if bad else 0

if __name__ == '__main__':
    sys.exit(main())
