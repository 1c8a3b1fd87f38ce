#!/bin/bash
# Effective shader clock and matrix-pipe occupancy of the encoder GEMMs (folded-LN forms, M = 124 096), persistent tile loop against
# one tile per workgroup, each looped on its own (one rocprofv3 --kernel-trace --pmc pass per configuration):
#   clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration;   busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r05}
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/pq_*
for shape in qkv proj fc1 fc2; do
  for persist in 0 1; do
    env GEMM_SHAPE=$shape PERSIST=$persist REPS=14 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d /tmp/pq_${shape}_$persist -o p -- python $R/tools/pmc_gemm_persist.py > /tmp/pq_${shape}_$persist.log 2>&1 || tail -3 /tmp/pq_${shape}_$persist.log
  done
done
TAG=$TAG python - <<'PY'
import collections, csv, glob, json, os
out = {}
for d in sorted(glob.glob("/tmp/pq_*")):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[3:]
    cf, kf = glob.glob(d + "/*counter_collection.csv"), glob.glob(d + "/*kernel_trace.csv")
    if not cf or not kf:
        continue
    dur = {}
    for r in csv.DictReader(open(kf[0])):
        dur[r.get("Dispatch_Id") or r.get("Dispatch_ID")] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    acc = collections.defaultdict(dict)
    for r in csv.DictReader(open(cf[0])):
        acc[r.get("Dispatch_Id") or r.get("Dispatch_ID")][r["Counter_Name"]] = float(r["Counter_Value"])
    rows = [(dur[d_][1], c["GRBM_GUI_ACTIVE"] / 8.0, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) for d_, c in acc.items()
            if d_ in dur and "gemm_pp64" in dur[d_][0] and dur[d_][1] > 0 and "GRBM_GUI_ACTIVE" in c]
    rows = rows[len(rows) // 3:]   # drop the first (cold-clock) third
    if rows:
        ns, cyc, busy = (sum(r[i] for r in rows) / len(rows) for i in range(3))
        shape, persist = tag.rsplit("_", 1)
        out[tag] = {"shape": shape, "persistent": bool(int(persist)), "dispatches": len(rows), "mean_us": ns / 1e3, "kcycles": cyc / 1e3,
                    "clock_GHz": cyc / ns, "mfma_busy_frac": busy / cyc, "busy_x_clock_GHz": busy / ns}
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/%s_pmc_clock_persist.json" % os.environ["TAG"], "w"), indent=1)
for k, v in out.items():
    print("%-8s persistent=%d  %7.1f us  %6.0f kcycles  clock %.3f GHz  matrix pipe busy %.3f  busy x clock %.3f GHz"
          % (v["shape"], v["persistent"], v["mean_us"], v["kcycles"], v["clock_GHz"], v["mfma_busy_frac"], v["busy_x_clock_GHz"]))
PY
