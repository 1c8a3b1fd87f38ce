#!/bin/bash
# tile-order A/B of the encoder GEMMs: time + FETCH_SIZE per order
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python tools/bench_gemm_order.py 464 468 470 472 476 2>&1 | grep "M=" | tee gpurun_out/r04j_gemm_order.txt
cd /tmp
for o in 464 468 470; do
  rm -rf /tmp/pmc_o$o
  ORDER_ONLY=$o timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_o$o -o p -- python $R/tools/bench_gemm_order.py > /tmp/pmc_o$o.log 2>&1 || tail -3 /tmp/pmc_o$o.log
  python - <<PY | tee -a $R/gpurun_out/r04j_gemm_order.txt
import csv, glob, collections
f = glob.glob("/tmp/pmc_o$o/*counter_collection.csv")
acc = collections.OrderedDict()
if f:
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_pp64" in r["Kernel_Name"]:
            acc.setdefault(r["Kernel_Name"][:90] + "|" + r["Grid_Size"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("order $o  FETCH_SIZE x2 = %8.1f MB  (%d launches)  %s" % (2 * sum(v) / len(v) / 1024.0, len(v), k))
PY
done
