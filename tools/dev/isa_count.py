#!/usr/bin/env python3
"""usage: tools/dev/isa_count.py /tmp/isa/NAME.s kernel-regex — static instruction counts of a kernel by issue class
(2-clock: v_fma/v_fmac/v_mul/v_add/v_sub f32, v_add/sub_u32, v_and/or/xor, v_mov, shifts per inst_rate2; 4-clock: the rest)."""
import re, sys, collections
txt = open(sys.argv[1]).read()
rx = sys.argv[2]
m = re.search(r'^(_Z\S*(?:%s)\S*):.*?\n(.*?)s_endpgm' % rx, txt, re.S | re.M)
body = m.group(2)
ops = collections.Counter()
for line in body.splitlines():
    t = line.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'):
        continue
    ops[re.sub(r'_e32$|_e64$|_dpp$|_sdwa$', '', t[0])] += 1
fast = re.compile(r'^v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|subrev_f32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|mov_b32|lshlrev_b32|lshrrev_b32|ashrrev_i32|pk_mul_f32|pk_fma_f32|pk_add_f32)$')
v = {k: n for k, n in ops.items() if k.startswith('v_') and not k.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane'))}
nf = sum(n for k, n in v.items() if fast.match(k))
print(m.group(1)[:70])
print('VALU', sum(v.values()), '2-clock', nf, '4-clock', sum(v.values()) - nf, '| SALU', sum(n for k, n in ops.items() if k.startswith('s_')),
      '| vmem', sum(n for k, n in ops.items() if k.startswith(('global_', 'scratch_', 'buffer_', 'flat_'))), 'lds', sum(n for k, n in ops.items() if k.startswith('ds_')))
print(' '.join('%s:%d' % (k, n) for k, n in sorted(v.items(), key=lambda x: -x[1])[:28]))
