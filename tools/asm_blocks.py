#!/usr/bin/env python3
"""Per-basic-block instruction statistics of one kernel in a hipcc -save-temps .s file.
usage: asm_blocks.py file.s mangled_kernel_prefix [min_instructions]"""
import re, sys, collections
src, name = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
s = open(src).read()
m = re.search(r'^' + re.escape(name) + r'[^\n]*:.*?\.end_amdhsa_kernel', s, re.S | re.M)
body = m.group(0)
open('/tmp/asm/cur.s', 'w').write(body)
lines = body.split('\n')
for k in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_next_free_sgpr', 'ScratchSize', '; Occupancy', '; NumVgprs', '; NumAgprs', '; TotalNumVgprs'):
    for l in lines:
        if k in l: print(l.strip())
blocks = []; cur = None
for i, l in enumerate(lines):
    if re.match(r'^\.LBB\d+_\d+:', l):
        cur = [l.split(':')[0], i, collections.Counter()]; blocks.append(cur)
    elif cur is not None:
        t = l.strip().split()
        if t and not t[0].startswith(('.', ';')): cur[2][t[0]] += 1
tot = collections.Counter()
for name, i, c in blocks:
    tot.update(c)
    n = sum(c.values())
    if n > minn:
        fm = c['v_fmac_f64_e32'] + c['v_fma_f64'] + c['v_mul_f64'] + c['v_add_f64']
        print(name, 'line', i + 1, 'n', n, 'f64', fm, 'acc_rd', c['v_accvgpr_read_b32'], 'acc_wr', c['v_accvgpr_write_b32'],
              'scr', sum(v for k, v in c.items() if k.startswith('scratch')), 'gl', c['global_load_dwordx2'], 'gs', c['global_store_dwordx2'],
              'ds', sum(v for k, v in c.items() if k.startswith('ds_')), 'wait', c['s_waitcnt'], 'mov64', c['v_mov_b64_e32'], 'lane', c['v_readlane_b32'] + c['v_writelane_b32'])
print('total', sum(tot.values()), 'accvgpr', tot['v_accvgpr_read_b32'] + tot['v_accvgpr_write_b32'], 'scratch', sum(v for k, v in tot.items() if k.startswith('scratch')))
