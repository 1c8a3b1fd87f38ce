#!/bin/bash
# rocprofv3 kernel statistics of one command on the GPU box, names shortened: `bash tools/prof_kernels.sh <tag> <filter-regex> -- cmd ...`
# -> gpurun_out/<tag>_kernel_stats.txt (Name[:70], Calls, AverageNs, MinNs, MaxNs of the kernels matching the filter)
tag=$1; filt=$2; shift 3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/prof_$tag
# (rocprofv3 itself starts in /tmp, the profiled command in the repository root: relative paths of the command work)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- bash -c 'cd "$0" && exec "$@"' "$R" "$@" > /tmp/prof_$tag.log 2>&1 || tail -5 /tmp/prof_$tag.log
cd $R
python - "$tag" "$filt" <<'PY'
import csv, glob, re, sys
tag, filt = sys.argv[1], sys.argv[2]
fs = sorted(glob.glob("/tmp/prof_%s/**/*kernel_stats.csv" % tag, recursive=True))
if not fs:
    sys.exit("prof_kernels.sh: no kernel_stats.csv under /tmp/prof_%s:\n%s" % (tag, open("/tmp/prof_%s.log" % tag).read()[-1500:]))
f = fs[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(filt, r["Name"])]
out = open("gpurun_out/%s_kernel_stats.txt" % tag, "w")
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    line = "%-70s calls %5s avg %10.1f us  min %9.1f  max %9.1f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
    print(line)
    out.write(line + "\n")
PY
