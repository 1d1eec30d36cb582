#!/bin/bash
# usage: tools/evidence.sh <tag>   (on the GPU box)  — the per-round evidence set:
#   gpurun_out/<tag>_bench.json          python bench.py (default flags)
#   gpurun_out/<tag>_kernel_stats.md     rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc_{fetch,write,tcc,tcp,sq_issue,sq_lanes}.md   one rocprofv3 --pmc pass each (no trace flags)
#   gpurun_out/<tag>_traffic_extend.json HBM / L2 bytes per extend launch derived from the passes (-> profiles/traffic_extend.json)
#   gpurun_out/<tag>_stage_counters.json per-kernel counters + the csrc hash they belong to (-> profiles/stage_counters.json, read by bench.py)
# (the counter passes run with fuse=0: extension and shadow rays of a depth in their own launches, so that every stage has its
#  own kernel name to attribute counters to; the bench line and the kernel trace run the default, fused form)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
pass() { # name counters...
  name=$1; shift
  d=$R/gpurun_out/${tag}_pmc_$name; rm -rf $d
  (cd $R && timeout 1200 rocprofv3 --pmc "$@" -d $d -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --set fuse=0 > $d.log 2>&1)
  db=$(find $d -name "*.db" | head -1)
  (cd $R && python profiles/summarize.py pmc $db > gpurun_out/${tag}_pmc_$name.md; head -6 gpurun_out/${tag}_pmc_$name.md)
  echo $db
}
# counter passes first: bench.py reads the traffic figures derived from them (same code, same box)
f=$(pass fetch FETCH_SIZE | tail -1)
w=$(pass write WRITE_SIZE | tail -1)
t=$(pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum | tail -1)
p=$(pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum | tail -1)
q=$(pass sq_issue SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY | tail -1)
l=$(pass sq_lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE | tail -1)
# VALU instruction classes (for the VALU-time model: FMA / MUL / ADD_F32 occupy a SIMD for 2 cycles, transcendentals 8, the rest 4)
m=$(pass mix SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU | tail -1)
# validation of the VALU-time model: a kernel that is nothing but independent v_fma_f32 at 8 waves per SIMD, under the same counters
(cd $R/profiles/micro && [ -x valu_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value valu_calib.hip -o valu_calib)
dc=$R/gpurun_out/${tag}_pmc_calib; rm -rf $dc
(cd $R && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d $dc -- profiles/micro/valu_calib > $dc.log 2>&1; cat $dc.log | tail -3)
cal=$(find $dc -name "*.db" | head -1)
(cd $R && python profiles/summarize.py pmc $cal > gpurun_out/${tag}_pmc_calib.md; head -8 gpurun_out/${tag}_pmc_calib.md)
export VALU_CALIB_DB=$cal
(cd $R && python profiles/summarize.py traffic 256 4 terrain_1002k "$tag" $f $w $t $p $q $l > gpurun_out/${tag}_traffic_extend.json; cp gpurun_out/${tag}_traffic_extend.json profiles/traffic_extend.json)
h=$(cd $R && python -c "import bench; print(bench.csrc_hash())")
(cd $R && python profiles/summarize.py stages 256 4 terrain_1002k "$tag" $h $f $w $t $q $l $m > gpurun_out/${tag}_stage_counters.json; cat gpurun_out/${tag}_stage_counters.json; cp gpurun_out/${tag}_stage_counters.json profiles/stage_counters.json)
cd $R && timeout 900 python bench.py --stage-rates > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 1500 gpurun_out/${tag}_bench.json
d=$R/gpurun_out/${tag}_stats; rm -rf $d
(cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $d -- python bench.py > $d.log 2>&1)
(cd $R && python profiles/summarize.py stats $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_kernel_stats.md; head -12 gpurun_out/${tag}_kernel_stats.md)
# config 4 (atrium: instanced, textured — the k_shade_pt<true> variant) under the same protocol: kernel stats + three PMC passes
if [ -n "$EVIDENCE_ATRIUM" ]; then
  apass() { # name counters...
    name=$1; shift
    d=$R/gpurun_out/${tag}_atrium_pmc_$name; rm -rf $d
    (cd $R && timeout 1200 rocprofv3 --pmc "$@" -d $d -- python bench.py --workload atrium --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $d.log 2>&1)
    (cd $R && python profiles/summarize.py pmc $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_atrium_pmc_$name.md; head -6 gpurun_out/${tag}_atrium_pmc_$name.md)
  }
  apass fetch FETCH_SIZE
  apass write WRITE_SIZE
  apass sq_lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
  d=$R/gpurun_out/${tag}_atrium_stats; rm -rf $d
  (cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $d -- python bench.py --workload atrium --no-cpu-baseline > $d.log 2>&1)
  (cd $R && python profiles/summarize.py stats $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_atrium_kernel_stats.md; head -8 gpurun_out/${tag}_atrium_kernel_stats.md)
  (cd $R && timeout 600 python bench.py --workload atrium --no-cpu-baseline --stage-rates > gpurun_out/${tag}_atrium_bench.json 2>/dev/null; tail -c 400 gpurun_out/${tag}_atrium_bench.json)
fi
