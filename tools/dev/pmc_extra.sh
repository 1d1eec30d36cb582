#!/bin/bash
# usage (GPU box): tools/dev/pmc_extra.sh <tag>  — two more counter passes of the bench: VALU instruction classes; VMEM / store-path
# back-pressure and instruction fetch (gpurun_out/<tag>_pmc_{mix,vmem}.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift
  d=$R/gpurun_out/${tag}_pmc_$name; rm -rf $d
  (cd $R && timeout 1200 rocprofv3 --pmc "$@" -d $d -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $d.log 2>&1)
  (cd $R && python profiles/summarize.py pmc $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_pmc_$name.md; grep -c . gpurun_out/${tag}_pmc_$name.md)
}
pass mix SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU
pass vmem SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_WAIT_INST_LDS
pass misc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
