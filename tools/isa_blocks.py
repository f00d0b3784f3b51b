"""Per-basic-block instruction mix of one kernel from hipcc --save-temps assembly.
usage: python tools/isa_blocks.py <file.s> <kernel-name-substring>"""
import re
import sys
import collections

s = open(sys.argv[1]).read()
pat = sys.argv[2]
funcs = re.split(r'\n(?=_Z[\w]+:)', s)
for f in funcs:
    name = f.split(':', 1)[0]
    if pat not in name:
        continue
    print('==', name)
    blocks = re.split(r'\n(?=\.LBB[\d_]+:)', f)
    for b in blocks:
        label = b.split(':', 1)[0].strip().split('\n')[-1]
        c = collections.Counter()
        for line in b.split('\n'):
            m = re.match(r'^\s+([a-z_0-9]+)\s', line)
            if not m:
                continue
            op = m.group(1)
            if op.startswith('v_pk'):
                c['v_pk'] += 1
            elif op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'):
                c['v_lane'] += 1
            elif op.startswith('v_accvgpr'):
                c['v_acc'] += 1
            elif op.startswith('v_mov'):
                c['v_mov'] += 1
            elif op.startswith('v_'):
                c['v_other'] += 1
            elif op.startswith('ds_'):
                c['ds'] += 1
            elif op.startswith('buffer_load') or op.startswith('global_load'):
                c['vmem_ld'] += 1
            elif op.startswith('buffer_store') or op.startswith('global_store'):
                c['vmem_st'] += 1
            elif op.startswith('scratch_'):
                c['scratch'] += 1
            elif op == 's_barrier':
                c['barrier'] += 1
            elif op == 's_waitcnt':
                c['waitcnt'] += 1
            elif op.startswith('s_cbranch') or op == 's_branch':
                c['branch'] += 1
            elif op.startswith('s_load') or op.startswith('s_buffer_load'):
                c['smem'] += 1
            elif op.startswith('s_'):
                c['salu'] += 1
        tot = sum(c.values())
        if tot >= 20:
            print(f'{label:12s} total {tot:5d}  ' + '  '.join(f'{k}={v}' for k, v in sorted(c.items())))
