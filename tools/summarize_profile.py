"""Condense rocprofv3 CSVs of tools/profile_round.sh into profiles/<tag>_*."""
import collections
import csv
import json
import os
import sys

# usage: summarize_profile.py <dir of rocprofv3 CSVs> <tag> [workload key [file suffix [command]]]
src, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "c3_300x300x128"
suffix = sys.argv[4] if len(sys.argv) > 4 else ""
command = sys.argv[5] if len(sys.argv) > 5 else ("python bench.py --steps 5 --warmup 1 --no-cpu --conv-iters 10 "
                                                 "--no-conv-beyond-mall")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "")


def per_kernel(path, counter=None):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        if counter and r["Counter_Name"] != counter:
            continue
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


stats = list(csv.DictReader(open(os.path.join(src, "trace_kernel_stats.csv"))))
fetch = per_kernel(os.path.join(src, "fetch_counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(os.path.join(src, "write_counter_collection.csv"), "WRITE_SIZE")
hit = per_kernel(os.path.join(src, "l2_counter_collection.csv"), "TCC_HIT_sum")
miss = per_kernel(os.path.join(src, "l2_counter_collection.csv"), "TCC_MISS_sum")

rows = []
traffic = {}
for s in stats:
    k = short(s["Name"])
    f = fetch.get(k)
    w = write.get(k)
    row = {"kernel": k, "calls": int(s["Calls"]), "avg_us": float(s["AverageNs"]) / 1e3,
           "min_us": float(s["MinNs"]) / 1e3, "max_us": float(s["MaxNs"]) / 1e3,
           "pct": float(s["Percentage"])}
    if f and w:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly half
        # of a wide coalesced stream (MI355X_MICROARCH.md "HBM") -> doubled.
        fb = 2.0 * 1024.0 * sum(f) / len(f)
        wb = 1024.0 * sum(w) / len(w)
        row.update(fetch_MB=fb / 1e6, write_MB=wb / 1e6, hbm_MB=(fb + wb) / 1e6)
        traffic[k] = fb + wb
    if hit.get(k) and miss.get(k):
        h, m = sum(hit[k]), sum(miss[k])
        row["l2_hit"] = h / (h + m) if h + m else None
    rows.append(row)

# k_mh_ws is one kernel per number of pending layers (sixth template argument): the
# colour classes of a sweep cycle through them, so the figure that compares with
# bench.py's launch average is the launch-weighted mean over the family.
families = collections.OrderedDict()
for r in rows:
    if r["kernel"].startswith("d3d::k_mh_ws<") and r["kernel"].count(",") >= 6:
        args = r["kernel"][len("d3d::k_mh_ws<"):-1].split(", ")
        args[5] = "*"                      # the pending-layer count
        families.setdefault("d3d::k_mh_ws<" + ", ".join(args) + ">", []).append(r)
variants = {}
for fam, members in families.items():
    calls = sum(m["calls"] for m in members)
    agg = {"kernel": fam, "calls": calls,
           "avg_us": sum(m["avg_us"] * m["calls"] for m in members) / calls,
           "min_us": min(m["min_us"] for m in members), "max_us": max(m["max_us"] for m in members),
           "pct": sum(m["pct"] for m in members)}
    if all("hbm_MB" in m for m in members):
        for key in ("fetch_MB", "write_MB", "hbm_MB"):
            agg[key] = sum(m[key] * m["calls"] for m in members) / calls
        for m in members:
            variants[m["kernel"]] = traffic.pop(m["kernel"])
        traffic[fam] = agg["hbm_MB"] * 1e6
    if all(m.get("l2_hit") is not None for m in members):
        agg["l2_hit"] = sum(m["l2_hit"] * m["calls"] for m in members) / calls
    rows.insert(rows.index(members[0]), agg)

with open(os.path.join(dst, "%s_kernel_stats%s.csv" % (tag, suffix)), "w") as fh:
    fh.write(open(os.path.join(src, "trace_kernel_stats.csv")).read())
with open(os.path.join(dst, "%s_summary%s.md" % (tag, suffix)), "w") as fh:
    fh.write("# rocprofv3 summary %s%s\n\n" % (tag, suffix))
    fh.write("command: `rocprofv3 --kernel-trace --stats -- %s` (+ separate `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, "
             "`--pmc TCC_HIT_sum TCC_MISS_sum` passes)\n\n" % command)
    fh.write("HBM bytes per launch = 2 x FETCH_SIZE KiB (gfx950 correction) + WRITE_SIZE KiB.  "
             "`k_mh_ws<..., *>` rows: launch-weighted mean over the kernel's pending-layer variants "
             "(sixth template argument; the seventh: 1/variance loads with the non-temporal hint), the figure bench.py's `avg_launch_us` compares with.\n\n")
    fh.write("| kernel | calls | avg us | min us | max us | % | fetch MB | write MB | HBM MB | L2 hit |\n")
    fh.write("|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write("| %s | %d | %.2f | %.2f | %.2f | %.2f | %s | %s | %s | %s |\n" % (
            r["kernel"], r["calls"], r["avg_us"], r["min_us"], r["max_us"], r["pct"],
            "%.1f" % r["fetch_MB"] if "fetch_MB" in r else "-",
            "%.1f" % r["write_MB"] if "write_MB" in r else "-",
            "%.1f" % r["hbm_MB"] if "hbm_MB" in r else "-",
            "%.3f" % r["l2_hit"] if r.get("l2_hit") is not None else "-"))
    for name in ("bench_plain.json", "bench_traced.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            fh.write("\n`%s`:\n\n```\n%s```\n" % (name, open(p).read()))
sys.path.insert(0, root)
from deconv3d_amd import _lib  # noqa: E402  (the library these counters were taken on)
json.dump({"tag": tag, "workload": workload, "source_hash": _lib.source_hash(),
           "hbm_bytes_per_launch": traffic,
           "hbm_bytes_per_launch_variants": variants},
          open(os.path.join(dst, "%s_traffic%s.json" % (tag, suffix)), "w"), indent=1)
print(open(os.path.join(dst, "%s_summary%s.md" % (tag, suffix))).read()[:3000])
