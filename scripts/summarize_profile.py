#!/usr/bin/env python3
"""rocprofv3 outputs of scripts/gpu_profile.sh (rocpd sqlite) -> the small summaries committed under profiles/:
per-kernel time statistics of the kernel-trace runs, per-kernel PMC counter sums, and profiles/<round>_traffic.json (HBM bytes
per alignment of the dominant fill kernel of each config, with the hash of the device sources they were measured on).
Run on the GPU box right after the passes (the databases are too big to travel); writes into gpurun_out/profiles_<round>/, to be
copied into profiles/.   usage: summarize_profile.py <round-name> <rocprof output dir>"""
import glob
import hashlib
import json
import os
import sqlite3
import sys

ROUND = sys.argv[1] if len(sys.argv) > 1 else "round5"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out", "profiles_" + ROUND)
os.makedirs(out_dir, exist_ok=True)


def dbs(pattern):
    return sorted(glob.glob(os.path.join(src, pattern, "**", "*results.db"), recursive=True))


for cfg in (2, 4, 5, 6):
    for d in dbs("trace_config%d" % cfg):
        db = sqlite3.connect(d)
        rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
                          "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        # round 6: launches of one kernel may OVERLAP (the pipelined fill series of ssw_host.c: two launches in flight share the compute units
        # -- a launch's duration is not its service time).  union_ns = the wall time during which
        # at least one dispatch of the kernel was running: for serial launches it equals total_ns, for a pipelined series it is the series' length.
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        union = {}
        if "start" in cols and "end" in cols:
            for name, in db.execute("select distinct name from kernels"):
                iv = sorted(db.execute("select start, end from kernels where name = ?", (name,)).fetchall())
                u = 0; cs, ce = None, None
                for a_, b_ in iv:
                    if cs is None: cs, ce = a_, b_
                    elif a_ <= ce: ce = max(ce, b_)
                    else: u += ce - cs; cs, ce = a_, b_
                if cs is not None: u += ce - cs
                union[name] = u
        with open(os.path.join(out_dir, ROUND + "_config%d_kernel_stats.csv" % cfg), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --config %d --steps 1 --warmup %d --cpu-sample 0 [--also none] --plain   (durations in ns; vgpr = the trace record's arch_vgpr field, NOT the allocation: the code objects say 72 for k_fill<10,frame>)\n" % (cfg, 0 if cfg == 5 else 1))
            f.write("# union_ns: wall time with at least one dispatch of the kernel running.  The fill launches of a bucket that needs several of them are PIPELINED since round 6 (main stream / a second\n")
            f.write("# stream alternately): their durations overlap, total_ns and avg_ns count waiting -- union_ns / passes over the batch is the fill time bench.py brackets with HIP events (roofline.launch_ms x launches).\n")
            f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent,vgpr,sgpr,lds_bytes,max_grid_x,workgroup_x,union_ns\n")
            for r in rows:
                f.write("\"%s\",%d,%d,%.0f,%d,%d,%.3f,%d,%d,%d,%d,%d,%d\n" % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], union.get(r[0], r[2])))
        print(open(os.path.join(out_dir, ROUND + "_config%d_kernel_stats.csv" % cfg)).read()[:1500])

import bench      # noqa: E402  (kernel_source_id: the hash over ssw_kernels.hip + lanes.h + ssw_dev.h that bench.py compares)
traffic = {"kernel_source_sha16": bench.kernel_source_id(),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (scripts/gpu_profile.sh); FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced read); KB -> bytes"}
# bench.py runs the batch `steps + warmup` times and once more (the upload-inclusive step of the DNA configs, the verification
# step of config 5): the PMC commands use --steps 1 --warmup 0, i.e. TWO passes of the fill kernel over the batch -- THREE for config 5,
# whose allocation warm-up is one chunk of DB entries = the whole reduced DB of the PMC command
PASSES = {2: 2, 4: 2, 5: 3}
alns = {2: 16000 * PASSES[2], 4: 10000 * PASSES[4], 5: 8192 * 2048 * PASSES[5]}
dominant = {2: "k_fill", 4: "k_chainq", 5: "k_filldb"}
with open(os.path.join(out_dir, ROUND + "_pmc.csv"), "w") as f:
    f.write("# rocprofv3 --pmc <counters> -- python bench.py --config N (config 2: 16000 reads, config 4: full size, config 5: 8192 x 2048), one pass per counter group\n")
    f.write("pass,kernel,counter,dispatches,sum,avg_per_dispatch,avg_dispatch_ns\n")
    sums = {}
    for d in dbs("pmc*"):
        name = [p for p in d.split(os.sep) if p.startswith("pmc")][0]
        c = sqlite3.connect(d)
        for r in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like '%k_%' group by kernel_name, counter_name order by kernel_name, counter_name"):
            f.write("%s,\"%s\",%s,%d,%.6g,%.6g,%.0f\n" % (name, r[0], r[1], r[2], r[3], r[4], r[5]))
            cfg = int(name[3])
            if dominant[cfg] in r[0] and not ("k_chainq" in r[0] and ", true," in r[0]) and r[1] in ("FETCH_SIZE", "WRITE_SIZE"):   # (window passes of the strip kernel excluded)
                sums.setdefault(cfg, {}).setdefault(r[1], 0.0)
                sums[cfg][r[1]] += r[3]
for cfg, v in sums.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        hbm = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        traffic["config%d" % cfg] = {"kernel": dominant[cfg] + "<...> (fill launches only; window passes of the same template excluded)", "alignments_counted": alns[cfg], "passes_over_the_batch": PASSES[cfg],
                                     "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"], "hbm_bytes_per_alignment": hbm / alns[cfg]}
if 2 in sums:
    traffic["config3"] = dict(traffic.get("config2", {}), note="config 3 shares config 2's kernel (k_fill<10,frame>); per-alignment traffic scales with the target length: not measured separately")
    traffic.pop("config3")
# VALU issue statistics of the dominant fill kernels from the SQ / GRBM passes just written (GRBM_GUI_ACTIVE sums the 8 XCDs)
import csv
rows = [r for r in csv.reader(l for l in open(os.path.join(out_dir, ROUND + "_pmc.csv")) if not l.startswith("#"))][1:]
issue = {}
for cfg, k in ((2, "k_fill<10, 3>"), (4, "k_chainq<12, false, 3>"), (5, "k_filldb<20, 16, true>")):
    d = {r[2]: (float(r[5]), float(r[6])) for r in rows if r[0].startswith("pmc%d" % cfg) and k in r[1]}
    if "SQ_INSTS_VALU" in d and "GRBM_GUI_ACTIVE" in d:
        valu, (gui, ns) = d["SQ_INSTS_VALU"][0], d["GRBM_GUI_ACTIVE"]
        issue["config%d" % cfg] = {"kernel": k, "SQ_INSTS_VALU_per_dispatch": valu, "GRBM_GUI_ACTIVE_per_dispatch": gui, "dispatch_ms": ns / 1e6,
                                  "effective_clock_GHz": round(gui / 8.0 / ns, 3), "cycles_per_valu_instruction": round(gui / 8.0 * 1024.0 / valu, 3),
                                  "issue_slot_occupancy_at_4_cycles": round(4.0 * valu / (gui / 8.0 * 1024.0), 3)}
# LDS / VALU utilisation per kernel (round 5: the counters the contract names).  Per kernel and pass the per-dispatch averages:
#   LDS bank-conflict rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS-array cycles / all LDS-array cycles, MI355X_MICROARCH.md LDS section)
#   VALU busy              = 4 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   (share of the 4-cycle issue slots; gfx94x's VALUBusy uses SQ_ACTIVE_INST_VALU the same way)
#   LDS busy               = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8)       (share of the cycles in which a CU's LDS array works)
util = {}
WATCH = {2: ["k_fill<10, 3>"], 4: ["k_chainq<12, false, 3>", "k_chainq<12, true, 3>", "k_chainq<8, true, 3>", "k_trace_wave<16>", "k_trace_wave<4>", "k_trace_wave<1>"], 5: ["k_filldb<20, 16, true>", "k_filldb<19, 16, true>"]}
for cfg, names in WATCH.items():
    for k in names:
        d = {}
        for r in rows:
            if r[0].startswith("pmc%d" % cfg) and k in r[1]:
                d[r[2]] = (float(r[4]), float(r[5]), float(r[6]), int(r[3]))      # sum, avg per dispatch, avg ns, dispatches
        if not d:
            continue
        e = {"dispatches_per_pass": d[next(iter(d))][3]}
        for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS",
                  "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if c in d:
                e[c] = d[c][0]
        gui = e.get("GRBM_GUI_ACTIVE")
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_rate"] = round(e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"], 4)
        if gui and "SQ_INSTS_VALU" in e:
            e["valu_busy_4cycle_slots"] = round(4.0 * e["SQ_INSTS_VALU"] / (1024.0 * gui / 8.0), 4)
        if gui and "SQ_LDS_IDX_ACTIVE" in e:
            e["lds_busy"] = round(e["SQ_LDS_IDX_ACTIVE"] / (256.0 * gui / 8.0), 4)
        if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_INST_ANY" in e:
            e["issue_stall_share_of_wave_cycles"] = round(e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"], 4)
        util["config%d %s" % (cfg, k)] = e
traffic["utilisation"] = util
traffic["utilisation_note"] = ("sums over all dispatches of the kernel in its PMC passes (separate passes per counter group: the same command every time); "
                               "lds_bank_conflict_rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; valu_busy = 4 SQ_INSTS_VALU / (1024 GRBM_GUI_ACTIVE / 8); "
                               "lds_busy = SQ_LDS_IDX_ACTIVE / (256 GRBM_GUI_ACTIVE / 8)")
with open(os.path.join(out_dir, ROUND + "_utilisation.txt"), "w") as f:
    f.write("# LDS / VALU utilisation of the hot kernels, rocprofv3 PMC passes of scripts/gpu_profile.sh (%s); kernel source %s\n" % (ROUND, traffic["kernel_source_sha16"]))
    f.write("%-44s %10s %10s %12s %12s\n" % ("kernel (config)", "VALU busy", "LDS busy", "LDS conflict", "issue stall"))
    for k, e in util.items():
        f.write("%-44s %10s %10s %12s %12s\n" % (k, e.get("valu_busy_4cycle_slots", "-"), e.get("lds_busy", "-"), e.get("lds_bank_conflict_rate", "-"), e.get("issue_stall_share_of_wave_cycles", "-")))
issue["note"] = ("SQ_INSTS_VALU and GRBM_GUI_ACTIVE (summed over the 8 XCDs) of the dominant fill kernel, separate PMC passes (<round>_pmc.csv): "
                 "SIMD-cycles per VALU instruction = 1024 SIMDs x GRBM_GUI_ACTIVE / 8 / SQ_INSTS_VALU; above 1.0 occupancy = some 2-cycle adds pair up")
traffic["valu_issue"] = issue
with open(os.path.join(out_dir, ROUND + "_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
print(json.dumps(traffic, indent=1))
