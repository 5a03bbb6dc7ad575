"""Per-kernel averages of the PMC passes written by tools/collect_traffic.sh -> gpurun_out/traffic_<workload>_<mode>.json
(copy into profiles/r03_traffic.json under the workload's key)."""
import collections, csv, glob, json, re, sys
root, workload, mode = sys.argv[1:4]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{root}/*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"])
        acc[(m.group(1) if m else r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c + ("_KiB" if c.endswith("_SIZE") else "")] = sum(v) / len(v)
        out[k]["launches_" + c] = len(v)
keep = {k: v for k, v in out.items() if any(t in k for t in ("rows", "ode_step", "dg_", "membrane", "update"))}
json.dump({workload: keep}, open(f"{root}.json", "w"), indent=1, sort_keys=True)
for k, v in keep.items():
    hbm = 2 * v.get("FETCH_SIZE_KiB", 0) + v.get("WRITE_SIZE_KiB", 0)
    hit = v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1)
    print(f"{k:28s} fetch {v.get('FETCH_SIZE_KiB', 0) / 1024:9.2f} MiB (x2 on gfx950)  write {v.get('WRITE_SIZE_KiB', 0) / 1024:9.2f} MiB  "
          f"HBM-side {hbm / 1024:9.2f} MiB  L2 hit {hit:.3f}")
