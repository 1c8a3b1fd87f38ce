#!/bin/bash
# PMC refresh for the current GEMM (schedules 2 and 7) and attention kernels: separate rocprofv3 --pmc passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pg_$tag /tmp/pa_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pg_$tag -o p -- python $R/tools/pmc_gemm.py > /tmp/pg.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pa_$tag -o p -- python $R/tools/pmc_attn.py > /tmp/pa.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for d in glob.glob("/tmp/pg_*") + glob.glob("/tmp/pa_*"):
    fs = glob.glob(d + "/*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        name = "gemm_pp64 (schedule 7)" if "gemm_pp64" in k else "gemm_kernel 256x256 K-stream (schedule 2)" if ("gemm_kernel" in k and "Li8ELi2ELi4" in k) else "attn_fwd_kernel<bf16,2>" if "attn_fwd" in k else None
        if name: acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, cn), v in acc.items():
        out[name][cn] = sum(v) / len(v)
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_busy.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
