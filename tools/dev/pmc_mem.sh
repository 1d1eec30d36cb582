#!/bin/bash
# usage (GPU box): tools/dev/pmc_mem.sh <tag> [scene] — counter passes of the memory path (TA / TCP / TCC / SQC / TD) over two
# serialised 64-spp frames (tools/dev/pc_frame.py): where a latency-bound kernel waits.  gpurun_out/<tag>_mem_<pass>.md
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; scene=${2:-terrain}
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift
  d=/tmp/${tag}_mem_$name; rm -rf $d
  (cd $R && timeout -s KILL 100 rocprofv3 --pmc "$@" -d $d -- python tools/dev/pc_frame.py 2 64 $scene > $R/gpurun_out/${tag}_mem_$name.log 2>&1)
  (cd $R && python profiles/summarize.py pmc $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_mem_$name.md; grep -c . gpurun_out/${tag}_mem_$name.md)
}
# (a pass may hold 2 TA / TD counters, 4 TCP / TCC counters, 8 SQ counters: more and rocprofv3 aborts — and then hangs)
pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass tcp1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum
pass tcp3 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
pass tcc1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc2 TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pass sqc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL SQ_IFETCH SQ_INSTS_SMEM SQ_WAVE_CYCLES
pass td TD_TD_BUSY_sum TD_TC_STALL_sum
