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
vals = {}
for g in "abcd":
    files = glob.glob(os.path.join(out, "pmc" + g, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    rows = [r for r in rows if "mpcb_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    sel = [r for r in rows if int(r["Dispatch_Id"]) == last]
    waves = int(sel[0]["Grid_Size"]) // int(sel[0]["Workgroup_Size"])
    lines.append("pass pmc%s: kernel %s  grid %s wg %s LDS %s scratch %s VGPR %s AGPR %s SGPR %s (last dispatch, whole grid | per wave)" % (
        g, sel[0]["Kernel_Name"].split("::")[-1], sel[0]["Grid_Size"], sel[0]["Workgroup_Size"], sel[0]["LDS_Block_Size"],
        sel[0]["Scratch_Size"], sel[0]["VGPR_Count"], sel[0]["Accum_VGPR_Count"], sel[0]["SGPR_Count"]))
    for r in sorted(sel, key=lambda r: r["Counter_Name"]):
        v = float(r["Counter_Value"]); vals[r["Counter_Name"]] = v
        lines.append("  %-28s %14.6g   per-wave %12.6g" % (r["Counter_Name"], v, v / waves))
if "FETCH_SIZE" in vals or "WRITE_SIZE" in vals:
    f = vals.get("FETCH_SIZE", 0.0); w = vals.get("WRITE_SIZE", 0.0)
    lines.append("")
    lines.append("HBM traffic per launch (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE under-reports wide coalesced reads 2x on gfx950):")
    lines.append("  FETCH_SIZE %.0f KiB = %.2f MB raw (%.2f MB with the 2x correction);  WRITE_SIZE %.0f KiB = %.2f MB" % (f, f * 1024 / 1e6, 2 * f * 1024 / 1e6, w, w * 1024 / 1e6))
    if bench:
        lines.append("  algorithmic bytes per launch (bench roofline): %s" % json.dumps(bench.get("roofline")))
if vals.get("SQ_WAVE_CYCLES"):
    wc = vals["SQ_WAVE_CYCLES"]
    lines.append("")
    lines.append("derived: VALU busy %.1f %% of wave cycles, LDS-instruction active %.1f %%, wait-any %.1f %%, wait-inst-LDS %.1f %%" % (
        100 * vals.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * vals.get("SQ_ACTIVE_INST_LDS", 0) / wc,
        100 * vals.get("SQ_WAIT_ANY", 0) / wc, 100 * vals.get("SQ_WAIT_INST_LDS", 0) / wc))
if bench and ("FETCH_SIZE" in vals and "WRITE_SIZE" in vals):
    # measured HBM bytes per launch, corrected as the guide prescribes; bench.py reports it as roofline.traffic
    json.dump({"workload": bench.get("config", {}).get("workload"), "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"],
               "bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0},
              open(os.path.join(out, "traffic.json"), "w"))
open(os.path.join(out, "pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
