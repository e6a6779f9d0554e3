"""Condense a tools/profile_round.sh output directory into kernel_stats.csv and pmc.txt (the files copied to profiles/)."""
import csv, glob, json, os, sys

out = sys.argv[1]
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    with open(stats[0]) as f, open(os.path.join(out, "kernel_stats.csv"), "w") as g:
        g.write(f.read())
lines = []
bench = {}
try:
    bench = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
except Exception:
    pass
if bench:
    lines.append("# bench line of the same command: %s %s, %.3f ms/step, kernel avg via HIP events %.3f ms" % (
        bench.get("value"), bench.get("unit"), bench.get("ms_per_step", float("nan")), bench.get("roofline", {}).get("kernel_ms_avg", float("nan"))))
# one solve = several launches (round 3: first attempt, second start, restoration pass — 1..4 by cfg.restoration / cfg.second_start):
# every launch of the LAST solve of the run gets its own block; sums over them are "per solve launch" figures
cfgb = bench.get("config", {})
n_launch = int(cfgb.get("kernel_launches_per_solve") or 0)
if not n_launch:                       # bench lines older than that field: three launches (first attempt, second start, restoration pass)
    n_launch = 3
allv = {}
for g in "abcde":
    files = glob.glob(os.path.join(out, "pmc" + g, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    rows = [r for r in rows if "mpcb_kernel" in r["Kernel_Name"]]
    all_ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))
    if len(all_ids) % n_launch:
        lines.append("# WARNING: %d solve-kernel dispatches are not a multiple of %d launches per solve" % (len(all_ids), n_launch))
    ids = all_ids[-n_launch:]
    for pos, did in enumerate(ids):
        sel = [r for r in rows if int(r["Dispatch_Id"]) == did]
        waves = int(sel[0]["Grid_Size"]) // int(sel[0]["Workgroup_Size"])
        short = "launch %d of %d: %s" % (pos + 1, len(ids), sel[0]["Kernel_Name"].split("::")[-1])
        vals = allv.setdefault(short, {})
        lines.append("pass pmc%s: %s  grid %s wg %s LDS %s scratch %s VGPR %s AGPR %s SGPR %s (whole grid | per launched wave)" % (
            g, short, sel[0]["Grid_Size"], sel[0]["Workgroup_Size"], sel[0]["LDS_Block_Size"],
            sel[0]["Scratch_Size"], sel[0]["VGPR_Count"], sel[0]["Accum_VGPR_Count"], sel[0]["SGPR_Count"]))
        for r in sorted(sel, key=lambda r: r["Counter_Name"]):
            v = float(r["Counter_Value"]); vals[r["Counter_Name"]] = v
            lines.append("  %-28s %14.6g   per-wave %12.6g" % (r["Counter_Name"], v, v / waves))
tot_f = sum(v.get("FETCH_SIZE", 0.0) for v in allv.values()); tot_w = sum(v.get("WRITE_SIZE", 0.0) for v in allv.values())
if tot_f or tot_w:
    lines.append("")
    lines.append("HBM traffic per solve = sum over its launches (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE under-reports wide coalesced reads 2x on gfx950):")
    for k, v in allv.items():
        lines.append("  %-60s FETCH_SIZE %8.0f KiB  WRITE_SIZE %8.0f KiB" % (k, v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)))
    lines.append("  total: FETCH %.2f MB raw (%.2f MB with the 2x correction), WRITE %.2f MB" % (tot_f * 1024 / 1e6, 2 * tot_f * 1024 / 1e6, tot_w * 1024 / 1e6))
    if bench:
        lines.append("  algorithmic bytes per launch (bench roofline): %s" % json.dumps(bench.get("roofline")))
for k, vals in allv.items():
    if vals.get("SQ_WAVE_CYCLES"):
        wc = vals["SQ_WAVE_CYCLES"]
        lines.append("")
        lines.append("derived, %s: VALU busy %.1f %% of wave cycles, LDS-instruction active %.1f %%, wait-any %.1f %%, wait-inst-LDS %.1f %%, LDS bank conflict cycles %.1f %% of LDS index cycles" % (
            k, 100 * vals.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * vals.get("SQ_ACTIVE_INST_LDS", 0) / wc,
            100 * vals.get("SQ_WAIT_ANY", 0) / wc, 100 * vals.get("SQ_WAIT_INST_LDS", 0) / wc,
            100 * vals.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, vals.get("SQ_LDS_IDX_ACTIVE", 0))))
fp64 = sum(64.0 * (v.get("SQ_INSTS_VALU_ADD_F64", 0.0) + v.get("SQ_INSTS_VALU_MUL_F64", 0.0) + v.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) + 128.0 * v.get("SQ_INSTS_VALU_FMA_F64", 0.0)
           for v in allv.values())
if fp64:
    lines.append("")
    lines.append("FP64 operations per solve launch (wave instructions x 64 lanes, FMA counted twice; all launches of the solve): %.4g flop" % fp64)
if bench and (tot_f or tot_w):
    # measured HBM bytes per solve (both launches), corrected as the guide prescribes; bench.py reports it as roofline.traffic (offline)
    wl = bench.get("config", {}).get("workload", "")
    first = next((v for k, v in allv.items() if k.startswith("launch 1 ") and v.get("SQ_WAVE_CYCLES")), {})
    wc = first.get("SQ_WAVE_CYCLES", 0.0)
    json.dump({"workload_key": wl.split(":")[0], "workload": wl, "round": sys.argv[2] if len(sys.argv) > 2 else None, "fetch_kib": tot_f, "write_kib": tot_w,
               "bytes_per_launch": (2 * tot_f + tot_w) * 1024.0, "fp64_flop_per_launch": fp64 or None, "launches_per_solve": n_launch,
               "settings": {"second_start": cfgb.get("second_start"), "restoration": cfgb.get("restoration"), "integrator": cfgb.get("integrator"),
                            "start_steer": cfgb.get("start_steer")},
               "valu_busy": (first.get("SQ_ACTIVE_INST_VALU", 0.0) / wc) if wc else None, "wait_any": (first.get("SQ_WAIT_ANY", 0.0) / wc) if wc else None},
              open(os.path.join(out, "traffic.json"), "w"))
open(os.path.join(out, "pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
