"""Copy the artefacts written by tools/collect_profiles.sh / collect_traffic.sh / pmc_*.sh (gpurun_out/) into profiles/
under their round names (ROUND below) and REGENERATE the numbers quoted in profiles/README.md from those files (the section
between the `<!-- rNN:begin -->` / `<!-- rNN:end -->` markers is written by this script, never by hand: text and data cannot
drift apart; earlier rounds' sections stay as they are)."""
import csv
import glob
import json
import os
import shutil

ROUND = os.environ.get("ROUND", "r04")
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
for wl, cmd in (("driver", "python3 bench.py --steps 20 --warmup 5` (the driver's command: three legs in one process -- config 2, the 995 k-tet mesh, the DG variants -- so a kernel's row mixes their launches; the per-leg files follow)"),
                ("config2", "python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-config3 --no-dg` (config 2 only: one population per kernel)"),
                ("config3", "python3 bench.py --workload config3 --steps 20 --warmup 5 --cpu-steps 0 --solve-steps 0 --no-dg`"),
                ("config2h", "python3 bench.py --workload config2h --steps 20 --warmup 5 --cpu-steps 0 --solve-steps 0 --no-dg --no-overlap`"),
                ("dg", "python3 bench.py --variant dg --steps 50 --warmup 5 --solve-steps 0`"),
                ("dg_config2h", "python3 bench.py --variant dg --workload config2h --steps 20 --warmup 3 --solve-steps 0`")):
    name = f"{ROUND}_{wl}_kernel_stats.csv"
    if not cp(last(f"{G}/{ROUND}/stats_{wl}/*/*_kernel_stats.csv"), name):
        continue
    picks = []
    for r in csv.DictReader(open(os.path.join(P, name))):
        n = r["Name"]
        if any(t in n for t in ("emi_rows", "knp_rows", "knp_membrane", "ode_step", "emi_membrane", "writeback", "dg_",
                                "cg_dir", "down_kernel", "up_kernel", "bi_spmv", "bi_update", "dense_kernel", "form_kernel")):
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
    if not cp(os.path.join(G, ROUND, f + ".json"), f"{ROUND}_{f}.json"):
        continue
    d = line(os.path.join(P, f"{ROUND}_{f}.json"))
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
    wm = d.get("with_solves_reference_ksp_min_it")
    if wm:
        bits.append(f"with ksp_min_it honoured {wm['ms_per_step']:.3f} ms/step ({wm['knp_iterations_avg']:.2f} BiCGStab)")
    wr = d.get("with_solves_reference_options")
    if wr:
        bits.append(f"with the reference's Krylov options {wr['ms_per_step']:.3f} ms/step ({wr['emi_iterations_avg']:.2f} CG on the "
                    f"preconditioned norm + {wr['knp_iterations_avg']:.2f} GMRES)")
    wx = d.get("with_solves_reference_tests_fastest_methods")
    if wx:
        bits.append(f"with the reference's stopping rules and BiCGStab {wx['ms_per_step']:.3f} ms/step "
                    f"({wx['emi_iterations_avg']:.2f} CG + {wx['knp_iterations_avg']:.2f} BiCGStab)")
    tw = d.get("reference_faithful_knp_assembled_twice")
    if tw:
        bits.append(f"A_knp assembled twice {tw['ms_per_step']:.4f} ms/step, {tw['value']:.3e} dofs/s")
    ag = (d.get("cpu_baseline") or {}).get("agreement")
    if ag:
        bits.append(f"GPU vs CPU port on the same trajectory: max relative difference {ag['max']:.1e}")
    mf = d.get("roofline_membrane_facet_kernel")
    if mf:
        bits.append(f"facet kernel alone {mf['avg_launch_us']:.1f} us" + (f", traffic {mf['traffic'] / 1e6:.2f} MB" if mf.get("traffic") else ""))
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
    rows.append((f"`{ROUND}_{f}.json`", f"`{cmd}`", "; ".join(bits)))

# ---- whole DG steps with the device solves (tools/dg_solves.py) ---------------------------------------------------
for f, cmd in (("dg_solves_config2", "python tools/dg_solves.py --workload config2 --steps 10 --warmup 2"),
               ("dg_solves_config2_continuous_aux_space", "KNPEMI_DG_AUX_UNSPLIT=1 KNPEMI_DG_PLAIN_AGGREGATION=1 KNPEMI_DG_AUX_SMOOTHED=1 python tools/dg_solves.py --workload config2 --steps 10 --warmup 2` (the continuous auxiliary space of round 2 with a smoothed prolongator; SpMV kernels, filter and sub-cycle as now)"),
               ("dg_solves_hex_r1", "python tools/dg_solves.py --workload hex_r1 --steps 10 --warmup 2"),
               ("dg_solves_config2h", "python tools/dg_solves.py --workload config2h --steps 10 --warmup 2 --solve-steps 5")):
    if not cp(os.path.join(G, ROUND, f + ".json"), f"{ROUND}_{f}.json"):
        continue
    ws = line(os.path.join(P, f"{ROUND}_{f}.json"))["with_solves"]
    rows.append((f"`{ROUND}_{f}.json`", f"`{cmd}" + ("" if cmd.endswith(")") else "`"),
                 f"whole DG steps with the device solves: {ws['ms_per_step']:.2f} ms/step, {ws['cg_iterations_per_step']:.1f} CG + "
                 f"{ws['bicgstab_iterations_per_step']:.1f} BiCGStab iterations per step, first step with the hierarchy set-up "
                 f"{ws['first_step_with_amg_setup_s']:.1f} s"))

# ---- HBM traffic ------------------------------------------------------------------------------------------------
# keyed [bench workload][kernel name as bench.py reports it]
traffic = {}
tfile = os.path.join(P, f"{ROUND}_traffic.json")
if os.path.exists(tfile):
    traffic = json.load(open(tfile))
name = {"emi_rows_v2": "emi_rows_kernel", "knp_rows_v2": "knp_rows_kernel", "emi_rows_hex_v2": "emi_rows_kernel",
        "knp_rows_hex_v2": "knp_rows_kernel", "dg_emi_hex_box_kernel": "dg_emi_hex_kernel", "dg_knp_hex_box_kernel": "dg_knp_hex_kernel"}
sources = [("config2", "config2", "cg", {}), ("config3", "config3", "cg", {}), ("config2h", "config2h", "cg", {}),
           ("config2", "r1", "dg_tetrahedron", {}), ("config3", "r2", "dg_tetrahedron", {}), ("config2h", "r2", "dg_hexahedron", {}),
           ("config2h", "r2", "dg_hexahedron_general", {"dg_emi_hex_kernel": "dg_emi_hex_kernel_general", "dg_knp_hex_kernel": "dg_knp_hex_kernel_general"})]
for wl, key, tag, ren in sources:
    f = os.path.join(G, f"traffic_{key}_{tag}.json")
    if os.path.exists(f):
        got = json.load(open(f))[key]
        for k, v in got.items():
            if tag.startswith("dg") and not k.startswith("dg_"):
                continue
            k2 = ren.get(name.get(k, k), name.get(k, k))
            traffic.setdefault(wl, {})[k2] = v
if traffic:
    json.dump(traffic, open(tfile, "w"), indent=1, sort_keys=True)
    bits = []
    for wl, ks in sorted(traffic.items()):
        for k, v in sorted(ks.items()):
            if "rows" in k or "membrane" in k or k.startswith("dg_emi") or k.startswith("dg_knp"):
                hbm = (2 * v.get("FETCH_SIZE_KiB", 0) + v.get("WRITE_SIZE_KiB", 0)) / 1024
                bits.append(f"{wl} {k}: 2 x {v.get('FETCH_SIZE_KiB', 0) / 1024:.1f} + {v.get('WRITE_SIZE_KiB', 0) / 1024:.1f} = {hbm:.1f} MiB")
    rows.append((f"`{ROUND}_traffic.json`", "`tools/collect_traffic.sh config2`, `... config3`, `... config2h` (CG path, `bench.py`), `... r1 dg`, `... r2 dg`, "
                 "`... r2 dg hexahedron`, `... r2 dg hexahedron general` (DG variant, `tools/dg_time.py`): FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum, "
                 "TCC_MISS_sum each in its own `rocprofv3 --kernel-trace --pmc` pass",
                 "HBM-side bytes per launch, `(2 FETCH_SIZE + WRITE_SIZE) KiB` (gfx950 correction of the guide): " + "; ".join(bits)))

# ---- PMC summaries ----------------------------------------------------------------------------------------------
for src, dst, what in ((f"{ROUND}/pmc_ode/summary.json", f"{ROUND}_pmc_ode_config2.json",
                        f"`tools/pmc_ode.sh config2 {ROUND}/pmc_ode`: counter passes of the bench trajectory"),
                       (f"{ROUND}/pmc_rows_c3/summary.json", f"{ROUND}_pmc_rows_config3.json", f"`tools/pmc_rows.sh config3 {ROUND}/pmc_rows_c3`"),
                       (f"{ROUND}/pmc_rows_c2h/summary.json", f"{ROUND}_pmc_rows_config2h.json",
                        f"`tools/pmc_rows.sh config2h {ROUND}/pmc_rows_c2h` (SQ_* cycle counters are in units of four cycles)"),
                       (f"{ROUND}/pmc_dg_hex/summary.json", f"{ROUND}_pmc_dg_config2h.json",
                        f"`DG_TIME_ARGS='--cell hexahedron' tools/pmc_dg.sh 2 {ROUND}/pmc_dg_hex` (box-mesh kernels of the DG variant on 165 888 hexahedra)")):
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
begin, end = f"<!-- {ROUND}:begin -->", f"<!-- {ROUND}:end -->"
sec = [begin, "", f"## Round {int(ROUND[1:])}", "",
       "Written by `tools/refresh_profiles.py` from the files beside it (collected with `tools/collect_profiles.sh`, "
       "`tools/collect_traffic.sh`, `tools/pmc_ode.sh`, `tools/pmc_rows.sh`, `tools/pmc_dg.sh` on the MI355X box): every number below "
       "is read from the file it stands next to.", "",
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
