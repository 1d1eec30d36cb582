#!/bin/bash
# usage: tools/dev/isa.sh NAME "<extra hipcc flags>" [kernel-name-regex] — device ISA of kernels.hip built with the extra flags into
# /tmp/isa/NAME.s, and the register / scratch / spill figures of the kernels matching the regex (default: the traversal kernels)
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/rendering-fw_amd/csrc
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$R/include -I$C -Wno-unused-function -Wno-unused-result \
  --offload-device-only -S $2 $C/kernels.hip -o /tmp/isa/$1.s 2>/dev/null || { echo "compile failed"; exit 1; }
python3 - "$1" "${3:-k_trace_fusedILb0|k_primary_packetILb0|k_shade_ptILb0|k_shade_ptILb1}" <<'PY'
import re,sys
name,rx=sys.argv[1],sys.argv[2]
txt=open('/tmp/isa/%s.s'%name).read()
# metadata blocks
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)(?=\n  - \.agpr_count|\Z)', txt, re.S):
    pass
for blk in txt.split('  - .agpr_count:')[1:]:
    nm=re.search(r'\.name:\s+(\S+)',blk)
    if not nm or not re.search(rx,nm.group(1)): continue
    g=lambda k:(re.search(r'\.%s:\s+(\S+)'%k,blk) or [None,None])[1]
    print(nm.group(1)[:60], 'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'spill_v',g('vgpr_spill_count'),'spill_s',g('sgpr_spill_count'),'scratch',g('private_segment_fixed_size'),'lds',g('group_segment_fixed_size'))
PY
