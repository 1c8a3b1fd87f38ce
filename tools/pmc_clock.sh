#!/bin/bash
# Effective shader clock and matrix-pipe occupancy of the attention variants and of the GEMM, each in a loop of its own
# (rocprofv3 --kernel-trace --pmc, one counter group; the kernel trace of the same pass gives the wall time of every dispatch):
#   clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration;   busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
run() {  # tag, env assignment, script
  rm -rf /tmp/pc_$1
  env $2 NIMG=64 REPS=12 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pc_$1 -o p -- python $R/tools/$3 > /tmp/pc_$1.log 2>&1 || tail -3 /tmp/pc_$1.log
}
if [ -n "$ONLY_GEMM" ]; then   # quick pass: the GEMM shapes only
run gemm7 GEMM_MODE=7 pmc_gemm.py
run gemm7_long "GEMM_MODE=7 GEMM_SHAPE=long" pmc_gemm.py
run conv "GEMM_MODE=0 GEMM_SHAPE=conv" pmc_gemm.py
else
run attn1 ATTN_MODE=1 pmc_attn.py
run attn2 ATTN_MODE=2 pmc_attn.py
run attn3 ATTN_MODE=3 pmc_attn.py
run gemm7 GEMM_MODE=7 pmc_gemm.py
run gemm7_fc1 "GEMM_MODE=7 GEMM_SHAPE=fc1" pmc_gemm.py
run conv "GEMM_MODE=0 GEMM_SHAPE=conv" pmc_gemm.py
run gemm7_long "GEMM_MODE=7 GEMM_SHAPE=long" pmc_gemm.py
run gemm_hipblaslt GEMM_MODE=-1 pmc_gemm.py
fi
python - <<'PY'
import collections, csv, glob, json, os
out = {}
for d in sorted(glob.glob("/tmp/pc_*")):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[3:]
    cf = glob.glob(d + "/*counter_collection.csv")
    kf = glob.glob(d + "/*kernel_trace.csv")
    if not cf or not kf:
        continue
    dur = {}
    for r in csv.DictReader(open(kf[0])):
        dur[r.get("Dispatch_Id") or r.get("Dispatch_ID")] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    acc = collections.defaultdict(dict)
    for r in csv.DictReader(open(cf[0])):
        did = r.get("Dispatch_Id") or r.get("Dispatch_ID")
        acc[did][r["Counter_Name"]] = float(r["Counter_Value"])
    rows = []
    for did, c in acc.items():
        name, ns = dur.get(did, ("?", 0.0))
        hot = "attn" in name or "gemm_pp64" in name or (tag == "gemm_hipblaslt" and ("Cijk" in name or "gemm" in name.lower()))
        if hot and ns > 0 and "GRBM_GUI_ACTIVE" in c:
            rows.append((ns, c["GRBM_GUI_ACTIVE"] / 8.0, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0))
    rows = rows[len(rows) // 3:]   # drop the first (cold-clock) third
    if rows:
        ns = sum(r[0] for r in rows) / len(rows)
        cyc = sum(r[1] for r in rows) / len(rows)
        busy = sum(r[2] for r in rows) / len(rows)
        out[tag] = {"dispatches": len(rows), "mean_us": ns / 1e3, "kcycles": cyc / 1e3, "clock_GHz": cyc / ns, "mfma_busy_frac": busy / cyc}
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/%s_pmc_clock.json" % os.environ.get("TAG", "r04"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
