"""Average PMC counter values per kernel over the passes written by tools/pmc_rows.sh."""
import collections, csv, glob, json, re, sys
root = sys.argv[1]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{root}/p*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+)(<[^(]*>)?\(", r["Kernel_Name"])
        acc[(m.group(1) if m else r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
keep = {k: v for k, v in out.items() if "rows" in k or "ode" in k or "dg_" in k}
json.dump(keep, open(f"{root}/summary.json", "w"), indent=1, sort_keys=True)
for k, v in keep.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:24s} {x:16.1f}")
