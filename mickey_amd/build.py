"""Build libmickey_hip.so (gfx950) in-tree with hipcc.

    python -m mickey_amd.build [--force] [--save-temps]

hipcc cross-compiles for gfx950 without a GPU.  Objects are rebuilt only when a source (or a header)
is newer; the shared library lands in mickey_amd/lib/ and travels with the repo snapshot to the GPU
box (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIBNAME = "libmickey_hip.so"
ARCH = "gfx950"
# per-file extra flags.  The attention kernel's softmax is VALU-bound: without NaN-honouring semantics fmaxf needs no
# canonicalising v_max and folds into v_max3 (-18 % VALU instructions); its inputs are finite by construction.
# NOT -fassociative-math: the 32 / 64-queries-per-wave instantiations must sum a row in the SAME order (pair i of a 32-pair
# batch is bit-identical to the same pair run alone, tests/test_bench_config_gpu.py); free to re-associate, hipcc ordered
# the row sums differently in the two -- and emitted 25 % more instructions.  -fno-slp-vectorize: left alone the row sums
# become v_pk_add_f32, which beside MFMAs costs more than the two scalar adds it replaces (858 vs 9xx TFLOP/s).
EXTRA_FLAGS = {"mk_attention.hip": ["-fno-honor-nans", "-fno-signed-zeros", "-fno-trapping-math", "-fno-slp-vectorize"] +
               (["-DMK_ATTN_ABLATIONS"] if os.environ.get("MK_ATTN_ABLATIONS") else []) +
               (["-DMK_ATTN_LP_DBG"] if os.environ.get("MK_ATTN_LP_DBG") else []),
               "mk_input.hip": ["-ffp-contract=off"],   # cv2-exact coordinates: (d + 0.5) * scale - 0.5 must not become an fma
               # the ping-pong GEMM's epilogues are VALU-bound (both waves of a SIMD drain while the matrix pipe idles): SLP packs the
               # (sum, sum of squares) pairs of the row statistics into v_pk_add_f32, which keeps the DPP steps of their 16-lane
               # reduction from folding into v_add_f32_dpp (v_mov_b32_dpp + v_pk_add_f32 instead: 3 issues for 2) -- without it
               # the producer epilogue needs 8 % fewer VALU cycles, the consumer 4 % (same IEEE operations, bit-identical)
               "mk_gemm_pp64.hip": ["-fno-slp-vectorize"] +
                                   (["-DMK_LN_ABL=%s" % os.environ["MK_LN_ABL"]] if os.environ.get("MK_LN_ABL") else []),
               "mk_gemm.hip": (["-DMK_PP64_ABLATIONS"] if os.environ.get("MK_PP64_ABLATIONS") else []) +
                              (["-DMK_GEMM_ABLATIONS"] if os.environ.get("MK_GEMM_ABLATIONS") else [])}


USAGE_JSON = os.path.join(OBJDIR, "resource_usage.json")
FLAGS_JSON = os.path.join(OBJDIR, "object_flags.json")   # {object: the flags it was built with}: part of the staleness check
_REMARK = re.compile(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|"
                     r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|Dynamic Stack):\s+(\S+)")


def _load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _load_usage():
    return _load_json(USAGE_JSON)


def _parse_resource_remarks(out):
    """-> ([{name, sgprs, vgprs, agprs, scratch, sgpr_spill, vgpr_spill, occupancy}], the compiler output without the remarks)"""
    kernels, rest, skip = [], [], 0
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "Occupancy [waves/SIMD]": "occupancy"}
    for line in out.splitlines():
        m = _REMARK.search(line)
        if m and "kernel-resource-usage" in line:
            k, v = m.group(1), m.group(2)
            if k == "Function Name":
                kernels.append({"name": v})
                skip = 2   # the two source-context lines clang prints under the first remark of a kernel
            elif k in keys and kernels:
                kernels[-1][keys[k]] = int(v)
            continue
        if skip and (line.lstrip().startswith("|") or re.match(r"\s*\d+ \|", line)):
            skip -= 1
            continue
        rest.append(line)
    return kernels, "\n".join(rest)


def resource_usage():
    """{source file: [per-kernel register / scratch / spill figures]} as recorded by the last build() of each source (a pure
    query: it never compiles; sources without an entry -- objects built before this report existed -- are simply absent,
    `python -m mickey_amd.build --force` refreshes everything).  Entries of sources that no longer exist are dropped."""
    srcs = {os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip"))}
    return {k: v for k, v in _load_usage().items() if k in srcs}


def source_hash():
    """Identity of the kernel sources (csrc/*.hip, csrc/*.hpp, include/*.h): profiles/ files record it so that bench.py can
    tell whether a committed counter file was collected from the kernels it is running (there is no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, save_temps=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if not srcs:
        raise RuntimeError("no .hip sources under %s" % CSRC)
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_time = _newest(hdrs) if hdrs else 0.0
    cc = hipcc()
    # -Rpass-analysis=kernel-resource-usage: per-kernel registers / scratch / spills, parsed into build/resource_usage.json
    # (tests/test_host_cpu.py asserts that no hot kernel spills: one register-hungry epilogue variant inlined into the GEMM
    # kernel once made hipcc spill the accumulators of every tile of every launch, +25 % on all GEMMs, profiles/README.md, round 2)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage",
             "-I", INCLUDE]
    jobs = []
    objs = []
    # an object is stale when a source / header is newer OR when it was built with other flags (probe builds driven by
    # environment variables -- MK_LN_ABL, MK_ATTN_ABLATIONS ... -- must not survive into the next plain build)
    built_with = _load_json(FLAGS_JSON)
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        fl = " ".join(flags + EXTRA_FLAGS.get(os.path.basename(s), []))
        if (force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time)
                or built_with.get(os.path.basename(o)) != fl):
            built_with[os.path.basename(o)] = fl
            cmd = [cc] + flags + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
            if save_temps:
                cmd.insert(1, "-save-temps=obj")
            jobs.append((s, cmd))

    def run(job):
        s, cmd = job
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=OBJDIR)
        return s, r.returncode, r.stdout

    failed = False
    have = {os.path.basename(x) for x in srcs}
    usage = {k: v for k, v in _load_usage().items() if k in have}   # entries of deleted / renamed sources go
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for s, rc, out in ex.map(run, jobs):
            if verbose:
                print("[hipcc] %s -> %s" % (os.path.basename(s), "ok" if rc == 0 else "FAILED"))
            kernels, out = _parse_resource_remarks(out)
            if rc == 0:
                usage[os.path.basename(s)] = kernels
            if out.strip() and (rc != 0 or verbose):
                print(out)
            failed |= rc != 0
    if jobs:
        with open(USAGE_JSON, "w") as f:
            json.dump(usage, f, indent=1, sort_keys=True)
    if failed:
        raise RuntimeError("hipcc failed")   # (FLAGS_JSON not updated: the failed objects stay stale)
    if jobs:
        with open(FLAGS_JSON, "w") as f:
            json.dump(built_with, f, indent=1, sort_keys=True)
    lib = lib_path()
    if jobs or not os.path.exists(lib) or os.path.getmtime(lib) < _newest(objs):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout)
            raise RuntimeError("link failed")
        if verbose:
            print("[link] %s" % lib)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
