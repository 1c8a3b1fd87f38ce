#!/bin/bash
# Round evidence for the build that is benchmarked (run on the GPU box through gpurun):
#   1. (last, once the traffic file exists) the default bench line -> gpurun_out/${TAG}_bench_b32.json
#   2. rocprofv3 --kernel-trace --stats of bench.py --steps 5   -> gpurun_out/${TAG}_bench_b32_kernel_stats.csv (+ the line under rocprof)
#   3. PMC passes of the same workload, one counter group per pass (TCC: FETCH_SIZE costs 3 of 4 slots, WRITE_SIZE 2;
#      never combined with other trace domains), post-processed per kernel ROLE -> gpurun_out/${TAG}_pmc_traffic.json
#      (FETCH_SIZE / WRITE_SIZE, read by bench.py when its source hash matches) and gpurun_out/${TAG}_pmc_busy.json
#      (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE)
# Copy what should be judged from gpurun_out/ to profiles/.
TAG=${TAG:-r05}
STEPS_BENCH=${STEPS_BENCH:-20}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $R/bench.py --steps 5 --warmup 2 --lean > /tmp/prof_$TAG.log 2>&1
grep "^{\"metric\"" /tmp/prof_$TAG.log | tail -1 > $R/gpurun_out/${TAG}_bench_b32_under_rocprof.json
cp $(ls /tmp/prof_$TAG/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_bench_b32_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$tag -o p -- python $R/bench.py --steps 1 --warmup 1 --lean --no-kernel-events > /tmp/pmc_$tag.log 2>&1 || tail -3 /tmp/pmc_$tag.log
done
TAG=$TAG python - <<'PY'
import collections, csv, glob, json, os, re, sys
R = os.environ["GRAFT_REPO_ROOT"]
TAG = os.environ["TAG"]
sys.path.insert(0, R)
from mickey_amd import build as B

def role(name):
    """kernel name (mangled or demangled) -> role key"""
    if "gemm_pp64_kernel" in name:
        m = re.search(r"gemm_pp64_kernelI\w+?Li(\d)ELi(\d)E", name) or re.search(r"gemm_pp64_kernel<[^,]+,\s*(\d)[^,]*,\s*(\d)", name)
        if not m:   # rocprofv3 prints some instantiations half-demangled ("gemm_pp64_kernel<bool _Accum, int, E, 0>"): the conv ones
            return "conv_gemm(+unparsed pp64 names)"
        amode, kind = int(m.group(1)), int(m.group(2))
        if amode == 1:
            return "conv_gemm"
        return {0: "gemm_plain(head linears)", 1: "encoder_gemm_consumer(qkv,fc1)", 2: "encoder_gemm_producer(proj,fc2,patch)",
                3: "encoder_gemm_producer_f32out"}[kind]
    for key, r in (("attn_", "attention"), ("layernorm", "layernorm"), ("lse_partial", "matcher_pass1"), ("dual_softmax_apply", "matcher_pass2"),
                   ("lse_split", "matcher_pass1"), ("split_apply", "matcher_pass2"), ("dsc_split", "matcher_planes"), ("lse_final", "matcher_merge"),
                   ("exprace", "sampler"), ("ransac_hyp", "hypotheses"), ("gemm_kernel", "gemm_128")):
        if key in name:
            return r
    return None

def collect(tag, counters):
    fs = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % tag)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not fs:
        return acc
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] in counters:
            k = role(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc

meta = {"source_hash": B.source_hash(), "batch": 32, "forwards": 2, "command": "bench.py --steps 1 --warmup 1 --lean --no-kernel-events (2 forwards)",
        "note": "rocprofv3 --kernel-trace --pmc <one group>, one pass per group; values are means per launch; FETCH_SIZE / WRITE_SIZE "
                "in KiB as reported (gfx950: a streaming read fetches 2x what FETCH_SIZE says, MI355X_MICROARCH.md)"}
traffic = dict(meta)
f, w = collect("FETCH_SIZE", ("FETCH_SIZE",)), collect("WRITE_SIZE", ("WRITE_SIZE",))
roles = {}
for k in sorted(set(f) | set(w)):
    fv, wv = f[k].get("FETCH_SIZE", []), w[k].get("WRITE_SIZE", [])
    roles[k] = {"launches": len(fv) or len(wv), "FETCH_SIZE_KiB_per_launch": sum(fv) / len(fv) if fv else None,
                "WRITE_SIZE_KiB_per_launch": sum(wv) / len(wv) if wv else None}
# HBM-side bytes of the role per FORWARD: (f x FETCH_SIZE + WRITE_SIZE) KiB x launches / forwards.  f = 2 is the guide's
# calibration for 16-B/lane streaming reads (every matrix / row kernel here); the sampler reads p with 4-byte loads and IS a
# calibration case of its own: its histogram pass reads ncell x 4 B per pair exactly once (481 MB per 32 pairs) and the raw
# FETCH_SIZE of the whole stage is 516 MB (that read + the 30-MB block maxima + the candidates' cells), so f = 1 there.
FETCH_FACTOR = {"sampler": 1.0}
for k, v in roles.items():
    if v["FETCH_SIZE_KiB_per_launch"] is not None and v["WRITE_SIZE_KiB_per_launch"] is not None:
        v["fetch_factor"] = FETCH_FACTOR.get(k, 2.0)
        v["bytes_per_forward"] = (v["fetch_factor"] * v["FETCH_SIZE_KiB_per_launch"] + v["WRITE_SIZE_KiB_per_launch"]) * 1024.0 * v["launches"] / meta["forwards"]
traffic["roles"] = roles
enc = [k for k in roles if k.startswith("encoder_gemm")]
if enc:
    n = sum(roles[k]["launches"] for k in enc)
    traffic["encoder_gemm"] = {"launches": n,
                               "FETCH_SIZE_KiB_per_launch": sum(roles[k]["FETCH_SIZE_KiB_per_launch"] * roles[k]["launches"] for k in enc) / n,
                               "WRITE_SIZE_KiB_per_launch": sum(roles[k]["WRITE_SIZE_KiB_per_launch"] * roles[k]["launches"] for k in enc) / n}
json.dump(traffic, open("%s/gpurun_out/%s_pmc_traffic.json" % (R, TAG), "w"), indent=1)
busy = dict(meta)
b = collect("SQ_VALU_MFMA_BUSY_CYCLES", ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"))
busy["roles"] = {}
for k, d in sorted(b.items()):
    m, gui = d.get("SQ_VALU_MFMA_BUSY_CYCLES", []), d.get("GRBM_GUI_ACTIVE", [])
    if not m or not gui:
        continue
    mm, gg = sum(m) / len(m), sum(gui) / len(gui)
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (profiles/r01_pmc_busy.md)
    busy["roles"][k] = {"launches": len(m), "SQ_VALU_MFMA_BUSY_CYCLES": mm, "GRBM_GUI_ACTIVE": gg,
                        "mfma_busy_frac": (mm / 1024.0) / (gg / 8.0)}
json.dump(busy, open("%s/gpurun_out/%s_pmc_busy.json" % (R, TAG), "w"), indent=1)
print(json.dumps({"traffic_roles": {k: v for k, v in roles.items() if "gemm" in k or "attention" in k}, "busy": busy["roles"]}, indent=1)[:3000])
PY
# the bench line LAST, with the traffic file of this very build in place (bench.py reads profiles/pmc_traffic.json when the
# source hash matches), so that the line carries roofline.traffic and the per-stage traffic
cd $R
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json   # (the file name bench.py reads: TRAFFIC_SOURCE)
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py --steps $STEPS_BENCH --warmup 5 --detail gpurun_out/${TAG}_bench_detail.json 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_b32.json
fi
head -c 900 $R/gpurun_out/${TAG}_bench_b32.json; echo; head -6 $R/gpurun_out/${TAG}_bench_b32_kernel_stats.csv | cut -c1-200
