#!/bin/bash
# usage: BENCH_ARGS="..." tools/lanes.sh <tag>  — VALU lanes per instruction and instruction counts per kernel (one PMC pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
tools/pmc_pass.sh $1 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" > /dev/null
python - <<PY
import re
rows={}
for l in open("$R/gpurun_out/pmc_$1_1.md"):
    m=l.split("|")
    if len(m)<7 or m[1].strip() in ("kernel","---"): continue
    rows.setdefault(m[1].strip(),{})[m[2].strip()]=(float(m[4]),float(m[6]),int(m[3]))
for k,v in rows.items():
    if "SQ_ACTIVE_INST_VALU" in v and v["SQ_ACTIVE_INST_VALU"][0]>1e6:
        print("%-70s n=%3d lanes %.1f  insts %.0f M  us %.0f" % (k[:70], v["SQ_ACTIVE_INST_VALU"][2], v["SQ_THREAD_CYCLES_VALU"][0]/v["SQ_ACTIVE_INST_VALU"][0], v.get("SQ_INSTS_VALU",(0,))[0]/1e6, v["SQ_ACTIVE_INST_VALU"][1]))
PY
