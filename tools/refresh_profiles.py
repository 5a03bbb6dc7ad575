"""Copy the artefacts written by tools/collect_profiles.sh / collect_traffic.sh / pmc_*.sh (gpurun_out/) into profiles/
under their round-3 names and REGENERATE the numbers quoted in profiles/README.md from those files (the section between
the `<!-- r03:begin -->` / `<!-- r03:end -->` markers is written by this script, never by hand: text and data cannot drift
apart)."""
import csv
import glob
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def last(pattern):
    hits = sorted(glob.glob(pattern), key=os.path.getmtime)
    return hits[-1] if hits else None


def cp(a, b):
    """Fresh artefact -> profiles/; without one the file already in profiles/ (an earlier collection of this round) stays
    and keeps its row."""
    if a and os.path.exists(a):
        shutil.copy(a, os.path.join(P, b))
        return True
    return os.path.exists(os.path.join(P, b))


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


rows = []
# ---- kernel statistics ------------------------------------------------------------------------------------------
for wl, cmd in (("config2", "python3 bench.py --steps 20 --warmup 5` (the driver's command)"),
                ("config3", "python3 bench.py --workload config3 --steps 20 --warmup 5 --cpu-steps 0 --solve-steps 0 --no-dg`"),
                ("config2h", "python3 bench.py --workload config2h --steps 20 --warmup 5 --cpu-steps 0 --solve-steps 0 --no-dg --no-overlap`"),
                ("dg", "python3 bench.py --variant dg --steps 50 --warmup 5`")):
    name = f"r03_{wl}_kernel_stats.csv"
    if not cp(last(f"{G}/r03/stats_{wl}/*/*_kernel_stats.csv"), name):
        continue
    picks = []
    for r in csv.DictReader(open(os.path.join(P, name))):
        n = r["Name"]
        if any(t in n for t in ("emi_rows", "knp_rows", "knp_membrane", "ode_step", "emi_membrane", "writeback", "dg_",
                                "cg_dir", "down_kernel", "up_kernel", "bi_spmv", "bi_update", "dense_kernel")):
            short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            picks.append(f"`{short}` {float(r['AverageNs']) / 1e3:.1f} us x {r['Calls']}")
    rows.append((f"`{name}`", f"`rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}", "; ".join(picks[:14])))

# ---- bench lines ------------------------------------------------------------------------------------------------
for f, cmd in (("bench_config2", "python bench.py --steps 20 --warmup 5"),
               ("bench_config3", "python bench.py --workload config3 --steps 20 --warmup 5 --cpu-steps 0 --no-dg"),
               ("bench_config2h", "python bench.py --workload config2h --steps 20 --warmup 5 --cpu-steps 0 --no-dg"),
               ("bench_config5s", "python bench.py --workload config5s --steps 20 --warmup 5 --cpu-steps 0 --no-dg"),
               ("bench_r3", "python bench.py --workload r3 --steps 10 --warmup 5 --repeats 3 --cpu-steps 0 --solve-steps 0 --no-dg"),
               ("bench_dg_config3", "python bench.py --variant dg --workload config3 --steps 20 --warmup 3"),
               ("bench_dg_config2h", "python bench.py --variant dg --workload config2h --steps 20 --warmup 3 --solve-steps 0"),
               ("bench_dg_config2h_general_kernels", "KNPEMI_DG_HEX_GENERAL=1 python bench.py --variant dg --workload config2h --steps 20 --warmup 3 --solve-steps 0"),
               ("bench_config2_plain_solver_loops", "KNPEMI_NO_FUSED=1 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-dg --no-config3")):
    if not cp(os.path.join(G, "r03", f + ".json"), f"r03_{f}.json"):
        continue
    d = line(os.path.join(P, f"r03_{f}.json"))
    bits = [f"{d['ms_per_step']:.4f} ms/step, {d['value']:.3e} {d['unit']}"]
    t = d.get("timing")
    if t:
        bits.append(f"median of {len(t['repeats_ms_per_step'])} windows ({t['min_ms_per_step']:.4f}..{t['max_ms_per_step']:.4f})")
    ro = d.get("roofline") or {}
    if ro:
        bits.append(f"`roofline` {ro['kernel']} {ro['avg_launch_us']:.1f} us = {ro['frac']:.3f} of 8 TB/s"
                    + (f" ({ro['frac_by_counters']:.3f} by counters)" if ro.get("frac_by_counters") else ""))
    for k, v in (d.get("roofline_row_kernels_alone") or {}).items():
        bits.append(f"{k} alone {v['avg_launch_us']:.1f} us = {v['frac']:.3f}"
                    + (f" ({v['frac_by_counters']:.3f} by counters)" if v.get("frac_by_counters") else ""))
    if d.get("spike_window"):
        sw = d["spike_window"]
        bits.append(f"spike window {sw['ms_per_step']:.4f} ms/step (ODE sweep {sw['ode_kernel_us_per_step']:.0f} us avg, {sw['ode_kernel_us_max']:.0f} max)")
    ws = d.get("with_solves")
    if ws and "emi" in ws:
        bits.append(f"with_solves {ws['ms_per_step']:.3f} ms/step ({ws['emi']['iterations_avg']:.2f} CG + {ws['knp']['iterations_avg']:.2f} BiCGStab)")
    elif ws:
        bits.append(f"with_solves {ws['ms_per_step']:.3f} ms/step ({ws['cg_iterations_per_step']:.1f} CG + {ws['bicgstab_iterations_per_step']:.1f} BiCGStab)")
    rp = d.get("roofline_potential_kernel")
    if rp:
        bits.append(f"{rp['kernel']} {rp['avg_launch_us']:.1f} us = {rp['frac']:.3f}")
    cb = d.get("cpu_baseline")
    if cb:
        bits.append(f"CPU port {cb['value']:.2e} dofs/s on {cb['cores']} threads")
    leg = d.get("config3_leg")
    if leg:
        al = leg.get("roofline_row_kernels_alone") or {}
        bits.append("config3_leg " + f"{leg['ms_per_step']:.4f} ms/step; "
                    + ", ".join(f"{k} alone {v['avg_launch_us']:.1f} us = {v['frac']:.3f}" for k, v in al.items()))
    ku = d.get("kernels_us_per_step") or {}
    if ku:
        bits.append("kernels " + ", ".join(f"{k.replace('_kernel', '')} {v:.1f}" for k, v in ku.items()))
    rows.append((f"`r03_{f}.json`", f"`{cmd}`", "; ".join(bits)))

# ---- whole DG steps with the device solves (tools/dg_solves.py) ---------------------------------------------------
for f, cmd in (("dg_solves_config2", "python tools/dg_solves.py --workload config2 --steps 10 --warmup 2"),
               ("dg_solves_config2_continuous_aux_space", "KNPEMI_DG_AUX_UNSPLIT=1 KNPEMI_DG_PLAIN_AGGREGATION=1 KNPEMI_DG_AUX_SMOOTHED=1 python tools/dg_solves.py --workload config2 --steps 10 --warmup 2` (the continuous auxiliary space of round 2 with a smoothed prolongator; SpMV kernels, filter and sub-cycle as now)"),
               ("dg_solves_hex_r1", "python tools/dg_solves.py --workload hex_r1 --steps 10 --warmup 2"),
               ("dg_solves_config2h", "python tools/dg_solves.py --workload config2h --steps 10 --warmup 2 --solve-steps 5")):
    if not cp(os.path.join(G, "r03", f + ".json"), f"r03_{f}.json"):
        continue
    ws = line(os.path.join(P, f"r03_{f}.json"))["with_solves"]
    rows.append((f"`r03_{f}.json`", f"`{cmd}" + ("" if cmd.endswith(")") else "`"),
                 f"whole DG steps with the device solves: {ws['ms_per_step']:.2f} ms/step, {ws['cg_iterations_per_step']:.1f} CG + "
                 f"{ws['bicgstab_iterations_per_step']:.1f} BiCGStab iterations per step, first step with the hierarchy set-up "
                 f"{ws['first_step_with_amg_setup_s']:.1f} s"))

# ---- HBM traffic ------------------------------------------------------------------------------------------------
traffic = {}
name = {"emi_rows_v2": "emi_rows_kernel", "knp_rows_v2": "knp_rows_kernel"}
for wl in ("config2", "config3"):
    f = os.path.join(G, f"traffic_{wl}_cg.json")
    if os.path.exists(f):
        traffic[wl] = {name.get(k, k): v for k, v in json.load(open(f))[wl].items()}
if not traffic and os.path.exists(os.path.join(P, "r03_traffic.json")):
    traffic = json.load(open(os.path.join(P, "r03_traffic.json")))
if traffic:
    json.dump(traffic, open(os.path.join(P, "r03_traffic.json"), "w"), indent=1, sort_keys=True)
    bits = []
    for wl, ks in traffic.items():
        for k, v in ks.items():
            if "rows" in k or "membrane_kernel" in k:
                hbm = (2 * v.get("FETCH_SIZE_KiB", 0) + v.get("WRITE_SIZE_KiB", 0)) / 1024
                bits.append(f"{wl} {k}: 2 x {v.get('FETCH_SIZE_KiB', 0) / 1024:.1f} + {v.get('WRITE_SIZE_KiB', 0) / 1024:.1f} = {hbm:.1f} MiB")
    rows.append(("`r03_traffic.json`", "`tools/collect_traffic.sh config2`, `... config3` (FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum, "
                 "TCC_MISS_sum each in its own `rocprofv3 --kernel-trace --pmc` pass of `bench.py --steps 5 --warmup 2 --repeats 1`)",
                 "HBM-side bytes per launch, `(2 FETCH_SIZE + WRITE_SIZE) KiB` (gfx950 correction of the guide): " + "; ".join(bits)))

# ---- PMC summaries ----------------------------------------------------------------------------------------------
for src, dst, what in (("r03_pmc_ode/summary.json", "r03_pmc_ode_config2.json",
                        "`tools/pmc_ode.sh config2 r03_pmc_ode`: three counter passes of the bench trajectory, all three clean"),
                       ("r03_pmc_rows/summary.json", "r03_pmc_rows_config3.json", "`tools/pmc_rows.sh config3 r03_pmc_rows`")):
    if cp(os.path.join(G, src), dst):
        d = json.load(open(os.path.join(P, dst)))
        bits = []
        for k, v in d.items():
            if "SQ_WAVES" in v and "SQ_INSTS_VALU" in v:
                w = v["SQ_WAVES"]
                msg = f"{k}: {v['SQ_INSTS_VALU'] / w:.0f} VALU + {v.get('SQ_INSTS_SALU', 0) / w:.0f} SALU + {v.get('SQ_INSTS_LDS', 0) / w:.0f} LDS instructions per wave"
                if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
                    msg += f", waiting {100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.0f} % of the wave cycles"
                bits.append(msg)
        rows.append((f"`{dst}`", what, "; ".join(bits)))

# ---- README section ---------------------------------------------------------------------------------------------
readme = os.path.join(P, "README.md")
text = open(readme).read()
begin, end = "<!-- r03:begin -->", "<!-- r03:end -->"
sec = [begin, "", "## Round 3", "",
       "Written by `tools/refresh_profiles.py` from the files beside it (collected with `tools/collect_profiles.sh`, "
       "`tools/collect_traffic.sh`, `tools/pmc_ode.sh`, `tools/pmc_rows.sh` on the MI355X box): every number below is read from "
       "the file it stands next to.  (Collected before the round's last change to the hexahedral aggregation, commit "
       "\"aggregate_apart: a root needs the neighbours it would take to be free\": `r03_dg_solves_hex_r1.json`, "
       "`r03_dg_solves_config2h.json` and the `with_solves` entry of `r03_bench_config2h.json` are from the build before it; "
       "the r = 1 box measured 3.75 ms per step with 6.6 CG + 2.9 BiCGStab iterations afterwards, DESIGN section 3.7.)", "",
       "| file | command | what it shows |", "|---|---|---|"]
sec += [f"| {a} | {b} | {c} |" for a, b, c in rows]
sec += ["", end]
block = "\n".join(sec)
if begin in text:
    text = text[:text.index(begin)] + block + text[text.index(end) + len(end):]
else:
    text = text.rstrip("\n") + "\n\n" + block + "\n"
open(readme, "w").write(text)
for a, b, c in rows:
    print(a, "--", c[:300])
