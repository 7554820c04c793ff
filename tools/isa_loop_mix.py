#!/usr/bin/env python3
"""Dev tool: per-loop instruction mix of one device function, from `hipcc -S --offload-device-only` output.

usage: python3 tools/isa_loop_mix.py <asm file> <mangled function name> [max loop length]
Lists every innermost loop (a backward branch with no other backward branch inside) with its instruction classes:
fp64 = v_mul/add/fma_f64, valu = other vector ALU, salu, smem (scalar loads), lds, vmem, wait (s_waitcnt/s_nop), br.
"""
import collections, re, sys

def cls(op):
    if op.startswith(('v_mul_f64', 'v_add_f64', 'v_fma_f64')): return 'fp64'
    if op.startswith('v_'): return 'valu'
    if op.startswith(('s_waitcnt', 's_nop')): return 'wait'
    if op.startswith(('s_cbranch', 's_branch')): return 'br'
    if op.startswith(('s_load', 's_buffer')): return 'smem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): return 'vmem'
    return 'other'

def main():
    src = open(sys.argv[1]).read()
    name = sys.argv[2]
    maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
    body = re.split(r'\n(?=%s:)' % re.escape(name), src)[1].split('.Lfunc_end')[0]
    lines = body.split('\n')
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
    inner = [(a, b) for a, b in loops if not any(a <= c and d <= b and (c, d) != (a, b) for c, d in loops)]
    print('# %s: %d instructions, %d loops, %d innermost' % (name, sum(1 for l in lines if l.startswith('\t') and not l.startswith('\t.')), len(loops), len(inner)))
    print('# label  instr | fp64 valu salu smem lds vmem wait br | marks')
    for a, b in inner:
        if b - a > maxlen: continue
        c = collections.Counter(); marks = set()
        for l in lines[a:b + 1]:
            t = l.strip()
            if not t or t.startswith((';', '.')): continue
            op = t.split()[0]; c[cls(op)] += 1
            if op == 's_set_gpr_idx_on': marks.add('index-mode multiplies')
            if op == 'ds_read_b64': marks.add('ds_read_b64')
            if op == 'ds_read_b128': marks.add('ds_read_b128')
            if op.startswith('v_exp') or op.startswith('v_ldexp'): marks.add('exp')
            if op.startswith('v_rcp_f64'): marks.add('rcp')
            if 'dpp' in t: marks.add('dpp')
        n = sum(c.values())
        label = lines[a].split(':')[0]
        print('%-12s %5d | %4d %4d %4d %4d %3d %4d %4d %2d | %s' % (label, n, c['fp64'], c['valu'], c['salu'], c['smem'], c['lds'], c['vmem'], c['wait'], c['br'], ', '.join(sorted(marks))))

if __name__ == '__main__':
    main()
