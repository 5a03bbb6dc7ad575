"""Copy the artefacts written by tools/collect_profiles.sh / collect_traffic.sh (gpurun_out/) into profiles/ under their
round-2 names and print the headline numbers."""
import csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
last = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)[-1]
cp = lambda a, b: shutil.copy(a, os.path.join(P, b))
cp(last(G + "/r02/stats_config2/runc/*_kernel_stats.csv"), "r02_config2_kernel_stats.csv")
cp(last(G + "/r02/stats_config3/runc/*_kernel_stats.csv"), "r02_config3_kernel_stats.csv")
cp(last(G + "/r02/stats_dg/runc/*_kernel_stats.csv"), "r02_dg_config2_kernel_stats.csv")
cp(last(G + "/r02/stats_config2h/runc/*_kernel_stats.csv"), "r02_config2h_kernel_stats.csv")
for a, b in (("r02/bench_config2.json", "r02_bench_config2.json"), ("r02/bench_config3.json", "r02_bench_config3.json"),
             ("r02/bench_config5s.json", "r02_bench_config5s.json"), ("r02/bench_config2h.json", "r02_bench_config2h.json"), ("r02/stats_dg.json", "r02_bench_dg_config2.json"),
             ("r02/bench_dg_config3.json", "r02_bench_dg_config3.json")):
    cp(os.path.join(G, a), b)
name = {"emi_rows_v2": "emi_rows_kernel", "knp_rows_v2": "knp_rows_kernel"}
out = {}
for wl, f in (("config2", "traffic_config2_cg.json"), ("config3", "traffic_config3_cg.json")):
    out[wl] = {name.get(k, k): v for k, v in json.load(open(os.path.join(G, f)))[wl].items()}
for wl, f, key in (("config2", "traffic_r1_dg.json", "r1"), ("config3", "traffic_r2_dg.json", "r2")):
    for k, v in json.load(open(os.path.join(G, f)))[key].items():
        if k.startswith("dg_"):
            out[wl][k] = v
json.dump(out, open(os.path.join(P, "r02_traffic.json"), "w"), indent=1, sort_keys=True)
for f in ("r02_bench_config2", "r02_bench_config3", "r02_bench_config5s", "r02_bench_config2h", "r02_bench_dg_config2", "r02_bench_dg_config3"):
    d = json.loads(open(os.path.join(P, f + ".json")).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 4), "%.3e" % d["value"], round(d["roofline"]["frac"], 3), d["roofline"]["kernel"],
          round(d["roofline"]["avg_launch_us"], 1), (d.get("with_solves") or {}).get("ms_per_step"),
          (d.get("cpu_baseline") or {}).get("value"), {k: round(v, 1) for k, v in d.get("kernels_us_per_step", {}).items()})
for f in ("r02_config2_kernel_stats.csv", "r02_dg_config2_kernel_stats.csv", "r02_config3_kernel_stats.csv", "r02_config2h_kernel_stats.csv"):
    print(f)
    for r in csv.DictReader(open(os.path.join(P, f))):
        n = r["Name"]
        if any(t in n for t in ("emi_rows", "knp_rows", "knp_membrane", "ode_step", "emi_membrane", "writeback", "dg_")):
            short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            print("   %-44s calls %5s avg %8.1f us" % (short[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
