#!/usr/bin/env python3
"""Neuron + astrocyte + ECS with a pulsed ECS K+/Na+ source, driven by a YAML config.

Same structure, units (ms, cm, mM, mV), parameters and time loop as the reference's
`examples/local_astrocyte_depolarization/run_stim_duration.py:150-500`: three sub-domains (ECS tag 0, neuron
tag 1 with the HH-mV model, glia tag 2 with the Kir4.1 / pump model), tortuosity-scaled diffusion coefficients,
a source `f_value` in the box [x_L, x_U] x [y_L, y_U] x [z_L, z_U] that is on for `pulse_width` ms every
`period` ms after `delay` until `end_time`, results every `save_frequency` steps.

Differences (outside the hot path): the emimesh tetrahedral mesh of the reference is not shipped, so unless the
config names a `mesh_file` (XDMF + HDF5 with `cell_marker` / `facet_marker`, read with `knpemi.fem.XDMFFile`), `mesh`
selects the synthetic four-cell box of `knpemi.fem.make_mesh_3D` scaled to centimetres with cells 1,3 tagged neuron
and 2,4 glia; results are one compressed `.npz` per saved step (the ADIOS2 checkpoints' role) and, with `--xdmf`,
the reference's `results_sub_<tag>.xdmf` / `results_mem_<tag>.xdmf` time series.  The source is a nodal
P1 field on the ECS (the reference evaluates the UFL conditional at quadrature points: identical when the region
is a union of cells, a one-cell ramp otherwise).

    python run_stim_duration.py -c baseline [--steps N] [--device-resident] [--direct]
"""
import argparse
import os
import sys
import time

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-fenics-x_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))

from knpemi import (create_functions_emi, create_functions_knp, create_solver_emi, create_solver_knp,  # noqa: E402
                    emi_system, knp_system, set_initial_conditions, setup_membrane_model,
                    update_ode_variables, update_pde_variables)
from knpemi.fem import Constant, Function, extract_submesh, make_mesh_3D  # noqa: E402
from setup_problem import load_model  # noqa: E402

# physical parameters, run_stim_duration.py:215-243 (ms, cm, mM)
DT = 0.1
C_M = 1.0
TEMPERATURE = 307e3
FARADAY = 96500e3
RGAS = 8.315e3
D_NA, D_K, D_CL = 1.33e-8, 1.96e-8, 2.03e-8
PSI = FARADAY / (RGAS * TEMPERATURE)
INIT = {"K": (3.092970607490389, 124.13988964240784, 99.3100014897692),
        "Na": (144.60625137617149, 12.850454639128186, 15.775818906083778),
        "Cl": (133.62525154406637, 5.0, 5.203660274163705)}

DEFAULTS = dict(mesh=dict(kind="box3d", resolution_factor=0, cell_type="tetrahedron", length=2), fname="baseline",
                Tstop=300, f_value=97, period=10, delay=1.0, pulse_width=1, end_time=100, lambda_e=1.2,
                lambda_i=2.55, x_L=2100.0e-7, x_U=2900.0e-7, y_L=2100.0e-7, y_U=2900.0e-7, z_L=2100.0e-7,
                z_U=2500.0e-7, save_frequency=5)


def load_config(name_or_path):
    """`config_files/<name>.yml` (run_stim_duration.py:503-516) merged over the defaults above."""
    path = name_or_path if os.path.isfile(name_or_path) else os.path.join(HERE, "config_files", f"{name_or_path}.yml")
    with open(path) as fh:
        cfg = yaml.load(fh, Loader=yaml.FullLoader) or {}
    out = dict(DEFAULTS)
    out.update(cfg)
    out["mesh"] = {**DEFAULTS["mesh"], **(cfg.get("mesh") or {})}
    for key in ("period", "pulse_width", "Tstop", "save_frequency"):
        if not out[key] > 0:
            raise ValueError(f"config: {key} must be positive")
    return out


def read_mesh(cfg):
    m = cfg["mesh"]
    if cfg.get("mesh_file"):            # the reference's key: XDMF with cell_marker / facet_marker (:61-80)
        from knpemi.fem import XDMFFile
        path = cfg["mesh_file"] if os.path.isabs(cfg["mesh_file"]) else os.path.join(HERE, cfg["mesh_file"])
        with XDMFFile(None, path, "r") as xdmf:
            mesh = xdmf.read_mesh(ghost_mode=None)
            ct = xdmf.read_meshtags(mesh, name="cell_marker")
            ft = xdmf.read_meshtags(mesh, name="facet_marker")
        return mesh, ct, ft
    mesh, ct, ft = make_mesh_3D(int(m["resolution_factor"]), m["cell_type"], l=int(m["length"]),
                                axon_tags=(1, 2, 1, 2))
    mesh.x[:] = mesh.x * 100.0   # metres -> centimetres (the units of this example)
    if getattr(mesh, "uniform_cell", None) is not None:
        mesh.uniform_cell = mesh.uniform_cell * 100.0      # the grid cell the generator handed on, in the same units
    return mesh, ct, ft


def source_is_active(t, cfg):
    """run_stim_duration.py:318-331,485: pulse train (t - delay) % period < pulse_width inside [delay, end_time]."""
    return (cfg["delay"] <= t <= cfg["end_time"]) and ((t - cfg["delay"]) % cfg["period"] < cfg["pulse_width"])


def source_region(x, cfg):
    return ((x[:, 0] > cfg["x_L"]) & (x[:, 0] < cfg["x_U"]) & (x[:, 1] > cfg["y_L"]) & (x[:, 1] < cfg["y_U"])
            & (x[:, 2] > cfg["z_L"]) & (x[:, 2] < cfg["z_U"]))


class Problem:
    """Everything `solve_system` builds before the time loop (run_stim_duration.py:150-440)."""

    def __init__(self, cfg, mesh_data=None):
        """mesh_data: (mesh, ct, ft) of one rank of a cell partition (knpemi.fem.distributed.make_partitioned_astro), as
        DOLFINx hands every MPI rank its part of the mesh the reference reads (run_stim_duration.py:127-134)."""
        self.cfg = cfg
        self.mesh, self.ct, self.ft = mesh, ct, ft = mesh_data if mesh_data is not None else read_mesh(cfg)
        ECS = {"name": "ECS", "tag": 0}
        neuron = {"name": "neuron", "tag": 1, "membrane_tags": [1], "ode_models": {1: load_model("hh_mv")}}
        glial = {"name": "glial", "tag": 2, "membrane_tags": [2], "ode_models": {2: load_model("glial")}}
        self.subdomain_list = subs = {0: ECS, 1: neuron, 2: glial}
        for tag, sd in subs.items():
            sm, e2p, v2p, _, _ = extract_submesh(mesh, ct, tag)
            sd.update(mesh_sub=sm, sub_to_parent=e2p, sub_vertex_to_parent=v2p)
            if tag > 0:
                g, g2p, _, _, _ = extract_submesh(mesh, ft, sd["membrane_tags"])
                sd.update(mesh_mem=g, mem_to_parent=g2p)
        self.dt = DT
        le, li = cfg["lambda_e"], cfg["lambda_i"]
        rho = {"z": -1}
        for tag in subs:
            rho[tag] = Constant(subs[tag]["mesh_sub"], INIT["Na"][tag] + INIT["K"][tag] - INIT["Cl"][tag])
        self.physical_parameters = pp = {
            "dt": Constant(mesh, DT), "n_steps_ODE": Constant(mesh, DT), "F": Constant(mesh, FARADAY),
            "psi": Constant(mesh, PSI), "C_phi": Constant(mesh, C_M / DT), "C_M": Constant(mesh, C_M),
            "R": Constant(mesh, RGAS), "temperature": Constant(mesh, TEMPERATURE), "rho": rho}

        def per_sub(e, i):
            return {0: Constant(subs[0]["mesh_sub"], e), 1: Constant(subs[1]["mesh_sub"], i),
                    2: Constant(subs[2]["mesh_sub"], i)}

        def init(name):
            return {t: Constant(subs[t]["mesh_sub"], INIT[name][t]) for t in subs}

        # ECS source fields (nodal): +f on K, -f on Na inside the region while the pulse is on
        self.region = source_region(subs[0]["mesh_sub"].x, cfg)
        Na = {"c_init": init("Na"), "z": 1.0, "name": "Na", "D": per_sub(D_NA / le ** 2, D_NA / li ** 2)}
        K = {"c_init": init("K"), "z": 1.0, "name": "K", "D": per_sub(D_K / le ** 2, D_K / li ** 2)}
        Cl = {"c_init": init("Cl"), "z": -1.0, "name": "Cl", "D": per_sub(D_CL / le ** 2, D_CL / li ** 2)}
        self.ion_list = ions = [K, Cl, Na]   # the last ion is eliminated (run_stim_duration.py:362)
        self.phi, self.phi_M_prev = create_functions_emi(subs, degree=1)
        self.c, self.c_prev = create_functions_knp(subs, ions, degree=1)
        V0 = self.c_prev[0][0].function_space
        self.f_source_K = Function(V0, name="f_source_K")
        self.f_source_Na = Function(V0, name="f_source_Na")
        K["f_source"] = self.f_source_K
        Na["f_source"] = self.f_source_Na    # as in the reference: carried by the eliminated ion, hence unused
        set_initial_conditions(ions, subs, self.c_prev)
        self.stim_params = {"stimulus": {"stim_amplitude": 0.0}, "stimulus_locator": lambda x: (x[0] < 20e-4)}
        for tag in (1, 2):
            subs[tag]["mem_models"] = setup_membrane_model(
                self.stim_params, pp, subs[tag]["ode_models"], ft, self.phi_M_prev[tag].function_space, ions)
        self.a_emi, self.p_emi, self.L_emi = emi_system(mesh, ct, ft, pp, ions, subs, self.phi, self.phi_M_prev,
                                                        self.c_prev, DT)
        self.a_knp, self.p_knp, self.L_knp = knp_system(mesh, ct, ft, pp, ions, subs, self.phi, self.phi_M_prev,
                                                        self.c, self.c_prev, DT)
        self.entity_maps = [subs[0]["sub_to_parent"], subs[1]["sub_to_parent"], subs[1]["mem_to_parent"],
                            subs[2]["sub_to_parent"], subs[2]["mem_to_parent"]]
        self.set_source(0.0)

    def set_source(self, t):
        """Nodal source fields at time t; returns True when they changed."""
        amp = self.cfg["f_value"] if source_is_active(t, self.cfg) else 0.0
        new = np.where(self.region, amp, 0.0)
        changed = not np.array_equal(new, self.f_source_K.x._a)
        if changed:
            self.f_source_K.x.array[:] = new
            self.f_source_Na.x.array[:] = -new
        return changed


def solve_odes(p, k):
    """run_stim_duration.py:92-123"""
    for tag, subdomain in p.subdomain_list.items():
        if tag == 0:
            continue
        phi_M_prev_sub = p.phi_M_prev[tag]
        for mem_model in subdomain["mem_models"]:
            ode_model = mem_model["ode"]
            update_ode_variables(ode_model, p.c_prev, phi_M_prev_sub, p.ion_list, p.subdomain_list, p.mesh, p.ct,
                                 tag, k)
            ode_model.step_lsoda(dt=p.dt, stimulus=p.stim_params["stimulus"],
                                 stimulus_locator=p.stim_params["stimulus_locator"])
            ode_model.get_membrane_potential(phi_M_prev_sub)
            for ion, I_ch_k in mem_model["I_ch_k"].items():
                ode_model.get_parameter("I_ch_" + ion, I_ch_k)


class XdmfResults:
    """results_sub_<tag>.xdmf / results_mem_<tag>.xdmf time series as the reference writes them
    (run_stim_duration.py:36-90,442-463), through knpemi.fem.XDMFFile."""

    def __init__(self, p, outdir):
        from knpemi.fem import XDMFFile
        self.sub, self.mem = {}, {}
        for tag, sd in p.subdomain_list.items():
            self.sub[tag] = XDMFFile(None, os.path.join(outdir, f"results_sub_{tag}.xdmf"), "w")
            self.sub[tag].write_mesh(sd["mesh_sub"])
            if tag > 0:
                self.mem[tag] = XDMFFile(None, os.path.join(outdir, f"results_mem_{tag}.xdmf"), "w")
                self.mem[tag].write_mesh(sd["mesh_mem"])

    def write(self, p, t):
        for tag in p.subdomain_list:
            self.sub[tag].write_function(p.phi[tag], t)
            for f in p.c[tag]:
                self.sub[tag].write_function(f, t)
            if tag > 0:
                self.mem[tag].write_function(p.phi_M_prev[tag], t)

    def close(self):
        for f in list(self.sub.values()) + list(self.mem.values()):
            f.close()


def write_results(p, outdir, k, t):
    fields = {}
    for tag in p.subdomain_list:
        fields[f"phi_{tag}"] = p.phi[tag].x._a
        for ion, f in zip(p.ion_list[:-1], p.c[tag]):
            fields[f"{ion['name']}_{tag}"] = f.x._a
        fields[f"{p.ion_list[-1]['name']}_{tag}"] = p.ion_list[-1][f"c_{tag}"].x._a
        if tag > 0:
            fields[f"phi_M_{tag}"] = p.phi_M_prev[tag].x._a
    np.savez_compressed(os.path.join(outdir, f"step_{k:06d}.npz"), t=t, **fields)


def solve_system(config, n_steps=None, device_resident=False, direct=False, outdir=None, quiet=False, xdmf=False,
                 extrapolate_guess=True):
    p = Problem(config)
    n_total = int(round(config["Tstop"] / float(DT)))
    n_steps = n_total if n_steps is None else min(n_steps, n_total)
    if outdir is None:
        outdir = os.path.join(HERE, "results", str(config["fname"]))
    os.makedirs(outdir, exist_ok=True)
    xdmf_out = XdmfResults(p, outdir) if xdmf else None
    history = dict(t=[], source=[], phi_M_neuron=[], phi_M_glia=[], K_ecs_max=[], its_emi=[], its_knp=[])
    t = 0.0

    def record(its_e, its_k):
        history["t"].append(t)
        history["source"].append(float(p.f_source_K.x._a.max()))
        history["phi_M_neuron"].append(float(p.phi_M_prev[1].x._a.mean()))
        history["phi_M_glia"].append(float(p.phi_M_prev[2].x._a.mean()))
        history["K_ecs_max"].append(float(p.c_prev[0][0].x._a.max()))
        history["its_emi"].append(its_e)
        history["its_knp"].append(its_k)

    t_wall = time.perf_counter()
    if device_resident:
        # whole loop on the GPU: fields, tables, operators and the Krylov solves stay in HBM
        from knpemi import _lib as L
        from knpemi.stepper import DeviceStepper
        st = DeviceStepper((p.a_emi, p.p_emi, p.L_emi), (p.a_knp, p.p_knp, p.L_knp), p.c, p.c_prev, p.phi,
                           p.phi_M_prev, device_solves=(1e-6, 1e-7), extrapolate_guess=extrapolate_guess)
        for tag in (1, 2):
            for mm in p.subdomain_list[tag]["mem_models"]:
                st.add_membrane_model(mm["ode"], p.stim_params["stimulus"], p.stim_params["stimulus_locator"])
        st.set_source(0, p.f_source_K.x._a)
        for k in range(n_steps):
            st.step()
            t = t + DT
            if p.set_source(t):
                st.set_source(0, p.f_source_K.x._a)
            if (k % config["save_frequency"]) == 0 or k == n_steps - 1:
                st.download()
                e, kk = st.iterations[-2][1], st.iterations[-1][1]
                record(e, kk)
                write_results(p, outdir, k, t)
                if xdmf_out:
                    xdmf_out.write(p, t)
                if not quiet:
                    print(f"t = {t:.2f} ms  source {'on ' if history['source'][-1] else 'off'}  "
                          f"phi_M neuron {history['phi_M_neuron'][-1]:.4f} glia {history['phi_M_glia'][-1]:.4f} mV")
        st.dp.sync()
    else:
        problem_emi = create_solver_emi(p.a_emi, p.L_emi, p.phi, p.entity_maps, p.subdomain_list, None,
                                        direct=direct, p=p.p_emi, atol=1e-40, rtol=1e-6, threshold=0.9)
        problem_knp = create_solver_knp(p.a_knp, p.L_knp, p.c, p.entity_maps, p.subdomain_list, None,
                                        direct=direct, p=p.p_knp, atol=2e-40, rtol=1e-7, threshold=0.75)
        for k in range(n_steps):
            if not quiet:
                print(f"solving for t = {t:.2f} ms")
            solve_odes(p, k)
            problem_emi.solve()
            problem_knp.solve()
            update_pde_variables(p.c, p.c_prev, p.phi, p.phi_M_prev, p.physical_parameters, p.ion_list,
                                 p.subdomain_list, p.mesh, p.ct)
            t = t + DT
            p.set_source(t)
            if (k % config["save_frequency"]) == 0 or k == n_steps - 1:
                record(problem_emi.solver.getIterationNumber(), problem_knp.solver.getIterationNumber())
                write_results(p, outdir, k, t)
                if xdmf_out:
                    xdmf_out.write(p, t)
    if xdmf_out:
        xdmf_out.close()
    history["wall_s"] = time.perf_counter() - t_wall
    history["steps"] = n_steps
    return p, history


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", metavar="config", help="name in config_files/ (or a path to a .yml)", type=str,
                        default="baseline")
    parser.add_argument("--steps", type=int, default=None, help="stop after this many steps (default: Tstop / dt)")
    parser.add_argument("--device-resident", action="store_true", help="DeviceStepper + device Krylov solves")
    parser.add_argument("--direct", action="store_true", help="host LU solves (MUMPS stand-in)")
    parser.add_argument("--xdmf", action="store_true", help="also write results_sub_/results_mem_ XDMF time series")
    args = parser.parse_args()
    cfg = load_config(args.c)
    _, hist = solve_system(cfg, n_steps=args.steps, device_resident=args.device_resident, direct=args.direct,
                           xdmf=args.xdmf)
    print(f"{hist['steps']} steps in {hist['wall_s']:.2f} s; phi_M neuron {hist['phi_M_neuron'][-1]:.4f} mV, "
          f"glia {hist['phi_M_glia'][-1]:.4f} mV, max ECS K {hist['K_ecs_max'][-1]:.4f} mM")
