#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (what `rocprofv3 --kernel-trace --stats` / `--pmc X` write on this ROCm 7.2
image) into the small text summaries committed under profiles/.

  python profiles/summarize.py stats  gpurun_out/prof_stats/<host>/<pid>_results.db  > profiles/rNN_kernel_stats.md
  python profiles/summarize.py pmc    gpurun_out/prof_fetch/<host>/<pid>_results.db  > profiles/rNN_pmc_fetch.md
"""
import sqlite3
import sys


def stats(path):
    c = sqlite3.connect(path).cursor()
    print("| kernel | calls | total ms | avg us | % | vgpr | sgpr | lds B | scratch B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
        "max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    for r in rows:
        print("| `%s` | %d | %.3f | %.1f | %.1f | %d | %d | %d | %d | %d | %d |" % (
            r[0][:90], r[1], r[2] / 1e6, r[3] / 1e3, 100.0 * r[2] / total, r[4], r[5], r[6], r[7], r[8], r[9]))


def pmc(path):
    c = sqlite3.connect(path).cursor()
    print("| kernel | counter | dispatches | avg value | sum value | avg dispatch us |")
    print("|---|---|---|---|---|---|")
    rows = c.execute(
        "select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) from counters_collection "
        "group by kernel_name, counter_name order by sum(value) desc").fetchall()
    for r in rows:
        print("| `%s` | %s | %d | %.1f | %.1f | %.1f |" % (r[0][:90], r[1], r[2], r[3], r[4], r[5] / 1e3))


def traffic(spp, streams, workload, tag, *paths):
    """profiles/traffic_extend.json from the PMC passes of one evidence run (FETCH_SIZE, WRITE_SIZE and TCC_REQ_sum, each in
    its own rocprofv3 --pmc pass of the same bench command): per launch of the extend stage = the primary kernel
    k_primary_stream<false> (k_extend<1, false> before round 2g) + the bounce kernel k_trace_stream<false, false>."""
    import json
    tot, disp = {}, {}
    for path in paths:
        c = sqlite3.connect(path).cursor()
        for name, counter, n, total in c.execute(
                "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            if "k_extend<1, false>" in name or "k_primary_stream<false>" in name or "k_trace_stream<false, false>" in name:
                tot[counter] = tot.get(counter, 0.0) + total
                disp[counter] = disp.get(counter, 0) + n
    out = {"workload": workload, "spp": int(spp), "streams": int(streams)}
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        # gfx950: FETCH_SIZE reports half the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md §HBM; visible on k_resolve in
        # the same pass), WRITE_SIZE is exact; both in KB
        out["hbm_bytes_per_extend_launch"] = int((2.0 * tot["FETCH_SIZE"] / disp["FETCH_SIZE"] + tot["WRITE_SIZE"] / disp["WRITE_SIZE"]) * 1024)
    if "TCC_REQ_sum" in tot:
        out["l2_requests_per_extend_launch"] = int(tot["TCC_REQ_sum"] / disp["TCC_REQ_sum"])
        out["l2_bytes_per_extend_launch"] = int(tot["TCC_REQ_sum"] / disp["TCC_REQ_sum"] * 128)  # 128-byte lines
    for k in ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"):
        if k in tot:
            out[k.lower() + "_per_extend_launch"] = int(tot[k] / disp[k])
    if "SQ_THREAD_CYCLES_VALU" in tot and "SQ_ACTIVE_INST_VALU" in tot:
        # lanes active per VALU instruction (thread-cycles over instruction-cycles, both in 4-clock units)
        out["valu_lanes_active_extend"] = round(tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"], 2)
    if "SQ_WAIT_ANY" in tot and "SQ_WAVE_CYCLES" in tot:
        out["wave_cycles_waiting_frac_extend"] = round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 4)
    out["dispatches"] = disp
    out["provenance"] = ("%s: MI355X, separate rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                         "--no-roofline` (tools/evidence.sh); extend stage = k_primary_stream<false> + k_trace_stream<false,false> dispatches; "
                         "hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch, l2 bytes = TCC_REQ_sum x 128" % tag)
    print(json.dumps(out, indent=1))


def short_name(kernel):
    """`void rtk::k_trace_stream<true, false>(rtk::Params)` -> `k_trace_stream<true, false>`"""
    import re
    m = re.search(r"rtk::(k_\w+(?:<[^>]*>)?)", kernel)
    return m.group(1) if m else kernel


def stages(spp, streams, workload, tag, csrc_hash, *paths):
    """profiles/stage_counters.json: per kernel of the bench, averaged per dispatch, from the PMC passes of one evidence run
    (each counter set in its own rocprofv3 --pmc pass of the same command): HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, KB; gfx950
    correction as in traffic()), L2 bytes (TCC_REQ x 128), VALU wave-instructions, lanes per VALU instruction, dispatch
    duration under the (serialising) counter pass.  bench.py reads it for roofline.stages and drops it when csrc_hash differs
    from the sources it runs on."""
    import json
    tot, disp, dur = {}, {}, {}
    for path in paths:
        c = sqlite3.connect(path).cursor()
        for name, counter, n, total, d in c.execute(
                "select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
            k = short_name(name)
            tot.setdefault(k, {})[counter] = total
            disp.setdefault(k, {})[counter] = n
            dur.setdefault(k, []).append(d / 1e3)
    out = {"workload": workload, "spp": int(spp), "streams": int(streams), "tag": tag, "csrc_hash": csrc_hash, "kernels": {}}
    # VALU-time model: cycles a wave64 instruction occupies its SIMD by class — v_fma / v_mul / v_add_f32: 2 (what
    # profiles/micro/valu_calib.hip and tools/dev/micro/sdwa_micro.hip measure: 2.3-2.6 nominal clocks per instruction), transcendentals
    # 8 (v_rcp_f32 / v_sqrt_f32: 8.25 clocks, tools/dev/micro/inst_rate.hip; 16 until r04e), everything else — compares, selects, conversions, min / max, integer — 4.  SIMD cycles = GRBM_GUI_ACTIVE
    # (summed over the 8 XCDs) x 32 CUs x 4 SIMDs.  (Round 3's SQ_ACTIVE_INST_VALU x 4 exceeded 1: that counter equals SQ_INSTS_VALU here.)
    def model(t):
        if not (t.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_INSTS_VALU" in t and "SQ_INSTS_VALU_FMA_F32" in t):
            return None
        fast = t.get("SQ_INSTS_VALU_FMA_F32", 0.0) + t.get("SQ_INSTS_VALU_MUL_F32", 0.0) + t.get("SQ_INSTS_VALU_ADD_F32", 0.0)
        trans = t.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        slow = max(0.0, t["SQ_INSTS_VALU"] - fast - trans)
        cyc = t["GRBM_GUI_ACTIVE"] * 32.0 * 4.0
        # Round 5 (tools/dev/micro/inst_rate3.hip, mixes): the classes are not additive — a 4-clock instruction occupies its pipe for 4.3
        # clocks whatever it is mixed with, a 2-clock one issues in the gaps down to 2.5 clocks per instruction overall, v_rcp / v_sqrt
        # take 8.6 and do not overlap.  VALU time ~ max(4.3 x slow + 8.6 x trans, 2.5 x all); the additive figure stays beside it.
        # (Upper bounds both: the counters class only FMA / MUL / ADD_F32 as fast, not v_add_u32 / v_and / v_mov / v_lshrrev.)
        pipe = (4.3 * slow + 8.6 * trans) / cyc
        issue = 2.5 * t["SQ_INSTS_VALU"] / cyc
        return (2.0 * fast + 8.0 * trans + 4.0 * slow) / cyc, fast / t["SQ_INSTS_VALU"], trans / t["SQ_INSTS_VALU"], pipe, issue
    # validation: the model on a kernel that is nothing but independent v_fma_f32 at 8 waves per SIMD (must read ~1)
    import os
    calib = None
    cal_db = os.environ.get("VALU_CALIB_DB")
    if cal_db and os.path.exists(cal_db):
        ct = {}
        for name, counter, total in sqlite3.connect(cal_db).cursor().execute(
                "select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
            if "k_calib_fma" in name:
                ct[counter] = total
        ct.setdefault("SQ_INSTS_VALU_MUL_F32", 0.0), ct.setdefault("SQ_INSTS_VALU_ADD_F32", 0.0), ct.setdefault("SQ_INSTS_VALU_TRANS_F32", 0.0)
        mv = model(ct)
        if mv:
            calib = {"kernel": "k_calib_fma (profiles/micro/valu_calib.hip): independent v_fma_f32 only, 8 waves per SIMD, every CU",
                     "valu_busy_frac_by_the_model": round(mv[0], 4), "fma_share_of_valu_instructions": round(mv[1], 4),
                     "lanes_per_instruction": round(ct["SQ_THREAD_CYCLES_VALU"] / ct["SQ_ACTIVE_INST_VALU"], 2) if ct.get("SQ_ACTIVE_INST_VALU") else None}
    out["valu_busy_validation"] = calib
    for k, t in tot.items():
        if not k.startswith("k_"):
            continue
        def per(c):
            return t[c] / disp[k][c] if c in t else None
        e = {"dispatches": max(disp[k].values()), "avg_dispatch_us": round(sum(dur[k]) / len(dur[k]), 1)}
        if per("FETCH_SIZE") is not None and per("WRITE_SIZE") is not None:
            e["hbm_bytes_per_dispatch"] = int((2.0 * per("FETCH_SIZE") + per("WRITE_SIZE")) * 1024)
        if per("TCC_REQ_sum") is not None:
            e["l2_bytes_per_dispatch"] = int(per("TCC_REQ_sum") * 128)
        if per("SQ_INSTS_VALU") is not None:
            e["sq_insts_valu_per_dispatch"] = int(per("SQ_INSTS_VALU"))
        if "SQ_THREAD_CYCLES_VALU" in t and "SQ_ACTIVE_INST_VALU" in t and t["SQ_ACTIVE_INST_VALU"] > 0:
            e["valu_lanes_per_instruction"] = round(t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"], 2)
        # SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs of 32 CUs x 4 SIMDs
        if "SQ_ACTIVE_INST_VALU" in t and t.get("GRBM_GUI_ACTIVE", 0) > 0:
            mv = model(t)  # (needs the instruction-class pass and GRBM_GUI_ACTIVE of the same kernel)
            e["valu_busy_frac"] = round(mv[0], 4) if mv else None
            if mv:
                e["valu_2_cycle_share"] = round(mv[1], 4)
                e["valu_transcendental_share"] = round(mv[2], 4)
                e["valu_slow_pipe_frac"] = round(mv[3], 4)  # 4.3 clocks per 4-clock-class instruction, 8.6 per transcendental
                e["valu_issue_frac"] = round(mv[4], 4)      # 2.5 clocks per VALU instruction of any class
        # the CU's scalar unit: one scalar instruction per CU and clock, i.e. 4.46 clocks of a SIMD's turn each whatever it is
        # (tools/dev/micro/inst_rate5.hip, round 6): its share of the launch's SIMD cycles — beside valu_busy_frac, the other pipe a
        # wave-uniform ("packet") kernel can be bound by
        if per("SQ_INSTS_SALU") is not None:
            e["sq_insts_salu_per_dispatch"] = int(per("SQ_INSTS_SALU"))
            if t.get("GRBM_GUI_ACTIVE", 0) > 0:
                e["salu_busy_frac"] = round(4.46 * t["SQ_INSTS_SALU"] / disp[k]["SQ_INSTS_SALU"] * disp[k]["GRBM_GUI_ACTIVE"] / (t["GRBM_GUI_ACTIVE"] * 32.0 * 4.0), 4)
        if "SQ_WAIT_ANY" in t and "SQ_WAVE_CYCLES" in t and t["SQ_WAVE_CYCLES"] > 0:
            e["wave_cycles_waiting_frac"] = round(t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"], 4)
        out["kernels"][k] = e
    out["provenance"] = ("%s: MI355X, separate rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                         "--no-roofline` (tools/evidence.sh); per dispatch averages; hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, "
                         "l2 bytes = TCC_REQ_sum x 128; valu_busy_frac = (2 x (FMA + MUL + ADD_F32) + 8 x TRANS_F32 + 4 x the other VALU instructions) / "
                         "(GRBM_GUI_ACTIVE x 32 CUs x 4 SIMDs): the share of SIMD cycles a VALU instruction occupied, by instruction class "
                         "(valu_busy_validation: the same model on a pure v_fma_f32 kernel); valu_slow_pipe_frac / valu_issue_frac: round 5's non-additive model, "
                         "max(4.3 x slow + 8.6 x trans, 2.5 x all) / SIMD cycles (tools/dev/micro/inst_rate3.hip); salu_busy_frac = 4.46 x SQ_INSTS_SALU / SIMD cycles (tools/dev/micro/inst_rate5.hip); csrc_hash = bench.py csrc_hash() of the sources profiled" % tag)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:])
    elif sys.argv[1] == "stages":
        stages(*sys.argv[2:])
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
