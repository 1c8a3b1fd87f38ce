#!/bin/bash
# HBM traffic of the dominant kernel (encoder GEMM) inside the real forward: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (TCC has 4 slots: FETCH_SIZE 3 + WRITE_SIZE 2 do not fit together).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-alt --no-h2d --no-single > /tmp/pmc_$c.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % c)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != c: continue
        if "gemm_pp64_kernel" in k and "Li0E" in k: acc["gemm_dense_256"].append(float(r["Counter_Value"]))
        elif "attn_fwd" in k: acc["attn"].append(float(r["Counter_Value"]))
        elif "layernorm" in k: acc["layernorm"].append(float(r["Counter_Value"]))
    out[c] = {k: {"launches": len(v), "mean_KiB": sum(v) / len(v)} for k, v in acc.items()}
json.dump(out, open("%s/gpurun_out/r02_pmc_traffic.json" % __import__("os").environ["GRAFT_REPO_ROOT"], "w"), indent=1)
print(json.dumps(out))
PY
