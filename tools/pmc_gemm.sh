#!/bin/bash
# SQ counters of the GEMM schedules (separate rocprofv3 --pmc passes, kernel trace only), fc2 shape
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for mode in 7; do
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
  tag=m${mode}_$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pg_$tag
  GEMM_MODE=$mode timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pg_$tag -o p -- python $R/tools/pmc_gemm.py > /tmp/pg.log 2>&1 || tail -3 /tmp/pg.log
done
done
python - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for d in glob.glob("/tmp/pg_m*"):
    mode = os.path.basename(d).split("_")[1]
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for cn, v in acc.items():
            out[mode][cn] = sum(v) / len(v)
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2_pmc_gemm.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
