#!/usr/bin/env python3
"""usage: tools/dev/isa_live.py /tmp/isa/NAME.s kernel-regex [top]  — VGPR liveness of one kernel of a device-ISA listing
(tools/dev/isa.sh NAME "-gline-tables-only" keeps .loc lines): backward dataflow over the kernel's basic blocks; prints the
maximum number of live VGPRs, and the source lines (file:line from the .loc directives) around the points of highest pressure."""
import re, sys, collections
txt = open(sys.argv[1]).read()
rx = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
m = re.search(r'^(_Z\S*(?:%s)\S*):.*?\n(.*?)s_endpgm' % rx, txt, re.S | re.M)
files = dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt))
files.update(dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"([^"]+)"\s*$', txt, re.M)))
body = m.group(2).splitlines()
insts = []   # (op, operands-string, loc)
labels = {}
loc = None
for line in body:
    t = line.strip()
    if not t or t.startswith(';'):
        continue
    if t.startswith('.loc'):
        p = t.split()
        loc = (int(p[1]), int(p[2]))
        continue
    if t.startswith('.'):
        if t.endswith(':'):
            labels[t[:-1]] = len(insts)
        continue
    if t.endswith(':'):
        labels[t[:-1]] = len(insts)
        continue
    t = t.split(';')[0].strip()
    sp = t.split(None, 1)
    insts.append((sp[0], sp[1] if len(sp) > 1 else '', loc))

def vregs(opnd):
    out = []
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', opnd):
        out += list(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', opnd):
        out.append(int(a))
    return out

NODEF = re.compile(r'^(global_store|flat_store|scratch_store|buffer_store|ds_write|ds_store|v_cmp_|v_cmpx_|v_readlane|v_readfirstlane|global_atomic_(?!.*_rtn)|s_|exp|v_nop|ds_bpermute_dummy)')
ACC = re.compile(r'^(v_fmac|v_mac|v_writelane|v_pk_fmac|v_dot.*acc|v_mov_b32_dpp|v_mov_b32_sdwa)')
N = len(insts)
defs, uses, succ = [None] * N, [None] * N, [None] * N
for i, (op, opnd, _) in enumerate(insts):
    parts = [x.strip() for x in opnd.split(',')] if opnd else []
    d, u = [], []
    if op.startswith('s_') or not parts:
        pass
    elif NODEF.match(op) and not (op.startswith('global_atomic') and 'sc0' in opnd):
        for x in parts:
            u += vregs(x)
    else:
        d = vregs(parts[0])
        k = 1
        if op.startswith(('v_mad_u64', 'v_mad_i64', 'v_div_scale', 'v_add_co', 'v_sub_co', 'v_subrev_co', 'v_addc_co', 'v_subb_co')) and len(parts) > 1:
            k = 2
        for x in parts[k:]:
            u += vregs(x)
        if ACC.match(op) or 'dpp' in op or 'sdwa' in op:
            u += d
        if op == 'v_swap_b32':
            d = vregs(parts[0]) + vregs(parts[1]); u = d[:]
    defs[i], uses[i] = set(d), set(u)
    s = []
    if op == 's_branch':
        s = [labels.get(parts[0], None)]
    elif op.startswith('s_cbranch'):
        s = [labels.get(parts[0], None), i + 1]
    elif op.startswith('s_setpc') or op.startswith('s_endpgm'):
        s = []
    else:
        s = [i + 1]
    succ[i] = [x for x in s if x is not None and x < N]
live_in = [set() for _ in range(N)]
changed = True
while changed:
    changed = False
    for i in range(N - 1, -1, -1):
        out = set()
        for s in succ[i]:
            out |= live_in[s]
        new = (out - defs[i]) | uses[i]
        if new != live_in[i]:
            live_in[i] = new
            changed = True
press = [len(x) for x in live_in]
mx = max(press)
print(m.group(1)[:70], 'instructions', N, 'max live VGPRs', mx)
# pressure by source line: max pressure seen while executing instructions of that line
by = collections.defaultdict(int)
for i, (_, _, l) in enumerate(insts):
    if l:
        by[l] = max(by[l], press[i])
for (f, ln), p in sorted(by.items(), key=lambda x: -x[1])[:top]:
    print('%4d live  %s:%d' % (p, files.get(f, str(f)).split('/')[-1], ln))
# windows: stretch of instructions with pressure >= mx - 8
thr = mx - 8
i = 0
while i < N:
    if press[i] >= thr:
        j = i
        while j < N and press[j] >= thr - 4:
            j += 1
        locs = collections.Counter(insts[k][2] for k in range(i, j) if insts[k][2])
        print('window %d..%d (%d instr) peak %d lines:' % (i, j, j - i, max(press[i:j])), ' '.join('%s:%d' % (files.get(f, str(f)).split('/')[-1], ln) for (f, ln), _ in locs.most_common(12)))
        i = j
    else:
        i += 1
if len(sys.argv) > 4:
    print('live at entry:', sorted(live_in[0]))
    for i, (op, opnd, l) in enumerate(insts):
        if op.startswith('scratch_') or 'accvgpr' in op:
            print(i, op, opnd, 'live', press[i], l)
    i = press.index(mx)
    print('live set at peak', i, sorted(live_in[i]))
