#!/usr/bin/env python3
"""Op-class sequence of a kernel in the saved ISA (make -C muzero_amd/csrc asm): where the waits,
barriers, LDS-DMA requests and matrix blocks of the ring kernels fall.  usage: isa_steps.py <mangled-substring> [start] [len]"""
import re, sys
import glob
# every kernel unit of the library (mzsearch, mzsearch_f32v, mzsearch_bf16, mzsearch_x3w)
s = "\n".join(open(f).read() for f in sorted(glob.glob('build/mzsearch*-hip-amdgcn-amd-amdhsa-gfx950.s')))
pat = sys.argv[1]
m = [m for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):\s*(?:;.*)?\n', s, re.M)][0]
body = s[m.end():s.index('s_endpgm', m.end())].split('\n')
out = []
for ln in body:
    full = ln.strip()
    t = full.split(' ')[0].split('\t')[0]
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    if t.startswith('v_mfma'): k = 'MFMA'
    elif t.startswith('ds_read'): k = 'dsr'
    elif t.startswith('ds_write'): k = 'dsw'
    elif t.startswith('global_load_lds'): k = 'GLDS'
    elif t.startswith('global_load'): k = 'gld'
    elif t.startswith('flat_'): k = 'FLAT'
    elif t.startswith('global_store'): k = 'gst'
    elif t == 's_barrier': k = 'BARRIER'
    elif t == 's_waitcnt': k = 'W[' + full.replace('s_waitcnt ', '') + ']'
    elif t.startswith('s_load'): k = 'sld'
    elif t.startswith('s_cbranch') or t.startswith('s_branch'): k = 'BR'
    else: k = '.'
    if out and out[-1][0] == k: out[-1][1] += 1
    else: out.append([k, 1])
txt = ' '.join(f'{k}x{n}' if n > 1 else k for k, n in out)
a = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
print(txt[a:a + n])
