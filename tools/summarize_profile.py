#!/usr/bin/env python3
"""Turns a gpurun_out/prof_<tag>/ directory (written by tools/profile.sh on the GPU box) into the
committed evidence under profiles/: <tag>_kernel_stats.csv, <tag>_pmc.json, <tag>_summary.md and
traffic_<workload>.json (HBM bytes per launch of the dominant kernel, from the PMC counters,
corrected as MI355X_MICROARCH.md 'HBM' prescribes: FETCH_SIZE counts 64 B per 128-B request of a
wide coalesced stream on gfx950 -> x2; WRITE_SIZE is exact; both are in KiB)."""
import csv, glob, json, os, shutil, subprocess, sys, time

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
s = json.load(open(os.path.join(src, "summary.json")))
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:  # (the scratch directory may hold CSVs of earlier runs: the newest one belongs to this summary)
    shutil.copy(stats[-1], os.path.join(dst, tag + "_kernel_stats.csv"))
json.dump(s.get("pmc", {}), open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
bench = None
try:
    bench = json.loads([l for l in open(os.path.join(src, "bench_trace.json")) if l.startswith("{")][-1])
    json.dump(bench, open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w"), indent=1)
except Exception:
    pass
dom = [k for k in s.get("pmc", {}) if "loglik_stream_kernel<6" in k]
lines = ["# rocprofv3 summary `%s` (workload %s)" % (tag, workload), "",
         "Command: `tools/profile.sh %s` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --cpu-steps 0 --prewarm 100 --no-by-input`" % tag,
         "(mean from rocprofv3's stats CSV; median from the dispatch trace of the same run: the mean carries the first launches' outliers)",
         "plus separate `--pmc` passes (SQ / LDS / MFMA / FETCH_SIZE / WRITE_SIZE).", "",
         "| kernel | calls | avg µs | median µs | % of GPU time |", "|---|---|---|---|---|"]
med = s.get("kernel_medians", {})
for r in s.get("kernel_stats", []):
    mk = med.get(r["Name"], {})
    lines.append("| `%s` | %s | %.1f | %s | %s |" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                    "%.1f" % (mk["median_ns"] / 1e3) if mk else "-", r["Percentage"]))
if med:
    json.dump(med, open(os.path.join(dst, tag + "_kernel_medians.json"), "w"), indent=1, sort_keys=True)
if dom:
    c = s["pmc"][dom[0]]
    hbm = (c.get("FETCH_SIZE", 0) * 2 + c.get("WRITE_SIZE", 0)) * 1024
    cfg = (bench or {}).get("config", {})
    json.dump({"kernel": dom[0], "hbm_bytes_per_launch": hbm, "FETCH_SIZE_KiB": c.get("FETCH_SIZE"),
               "WRITE_SIZE_KiB": c.get("WRITE_SIZE"), "correction": "fetch x2 (gfx950 wide coalesced loads), write x1",
               "source": "profiles/%s_pmc.json" % tag,
               # the capture is valid for this workload only (bench.py emits `traffic` when these match its run)
               "draws": cfg.get("draws"), "tree": cfg.get("tree"), "nnz": cfg.get("nnz"),
               "source_id": (bench or {}).get("detail", {}).get("source_id"),
               # when and on which commit the counters were collected (bench.py copies both into roofline.traffic_source)
               "captured": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(os.path.getmtime(os.path.join(src, "summary.json")))),
               "commit": subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
                         + ("+dirty" if subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "polee_amd", "bench.py"], capture_output=True, text=True).stdout.strip() else "")},
              open(os.path.join(dst, "traffic_%s.json" % workload), "w"), indent=1)
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    lines += ["", "## Dominant kernel `%s`" % dom[0][:60], "",
              "* HBM traffic per launch (PMC): %.3f GB (FETCH_SIZE %.0f KiB x2 + WRITE_SIZE %.0f KiB)" % (hbm / 1e9, c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0)),
              "* shader cycles per launch (GRBM_GUI_ACTIVE/8): %.0f" % cyc,
              "* VALU busy: %.0f %% (SQ_ACTIVE_INST_VALU x4 / (cycles x 1024 SIMDs))" % (100 * c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(cyc * 1024, 1)),
              "* LDS busy: %.0f %% (SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs)); bank-conflict cycles %.2f %% of LDS cycles" % (
                  100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / max(cyc * 256, 1), 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
              "* waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES): %.0f %%" % (100 * c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)),
              "* matrix-core busy: %.0f %% (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)); %.1f M MFMA instructions" % (
                  100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(cyc * 1024, 1), c.get("SQ_INSTS_MFMA", 0) / 1e6),
              "* wave-level instructions: VALU %.1f M, LDS %.1f M, SALU %.1f M" % (c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_INSTS_LDS", 0) / 1e6, c.get("SQ_INSTS_SALU", 0) / 1e6)]
if bench:
    lines += ["", "bench.py line under the profiler (slower than an unprofiled run): `value` %.1f VI iters/s, dominant kernel %.3f ms" % (bench["value"], bench["roofline"]["kernel_ms_avg"])]
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
