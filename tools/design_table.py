"""Markdown of DESIGN.md section 5 from a bench DETAIL object (dev tool): python tools/design_table.py profiles/r05_bench_detail.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("`%s`: **%.1f pairs/s** (%.2f ms per step of %d pairs), dominant kernel %.0f %s = %.3f of peak; L2-miss-side traffic %.2f GB per "
      "launch against %.2f GB algorithmic.\n" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("pairs_per_step", 32),
                                                r["achieved"], r["unit"], r["frac"], (r["traffic"] or 0) / 1e9,
                                                r["algorithmic_bytes_per_launch"] / 1e9))
print("| stage | launches/step | ms/step | share | achieved | of peak | algorithmic MB/step | PMC traffic MB/step |")
print("|---|---|---|---|---|---|---|---|")
for s in r["stages"]:
    ach = "%.0f %s" % (s["achieved"], s["unit"]) if s.get("achieved") else "-"
    print("| %s | %d | %.2f | %.1f %% | %s | %s | %s | %s |" % (
        s["stage"], s["launches_per_step"], s["ms_per_step"], 100 * s["share_of_kernel_time"], ach,
        "%.2f" % s["frac"] if s.get("frac") else "-",
        "%.0f" % s["algorithmic_mb_per_step"] if s.get("algorithmic_mb_per_step") else "-",
        "%.0f" % s["traffic_mb_per_step"] if s.get("traffic_mb_per_step") else "-"))
print()
print("| leg | pairs/s | ms/step | what |")
print("|---|---|---|---|")
for k, v in (d.get("legs") or {}).items():
    if "value" in v:
        print("| %s | %.1f | %.2f | %s |" % (k, v["value"], v["ms_per_step"], v["what"][:110]))
sp, cb = d.get("single_pair") or {}, d.get("cpu_baseline") or {}
if sp and cb:
    print("\nSingle pair (configs[1], hipGraph replay): %.2f ms.  CPU oracle: %.3f pairs/s on %d of %d cores (%s)." % (
        sp["ms_per_pair"], cb["value"], cb["cores"], cb["cores_available"], cb["sample"][:80]))
for k in ("sustained", "pcie_inclusive", "precision"):
    if d.get(k):
        print(k + ":", json.dumps(d[k])[:1200])
