#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_WAIT_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|TCC_EA0_RDREQ[A-Z_0-9]*\|FETCH_SIZE\|WRITE_SIZE\|GRBM_GUI_ACTIVE\|SQ_INSTS_[A-Z_]*MFMA[A-Z_0-9]*\|TCP_[A-Z_]*TAGCONFLICT[A-Z_]*" $R/gpurun_out/counters.txt | sort -u | tr '\n' ' '
echo
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$name -o p -- python $R/tools/pmc_gemm.py > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== $set -> $f"
  python - "$f" <<'PY'
import sys, csv, collections
f=sys.argv[1]
if not f: sys.exit(0)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r.get('Kernel_Name','')
    if 'gemm' not in k: continue
    short='pp' if 'gemm_pp' in k else ('t2' if 'Li8ELi2ELi4' in k else 't1')
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
done
