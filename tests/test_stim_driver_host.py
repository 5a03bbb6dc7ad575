"""CPU tests of the YAML-configured multi-tag driver (examples/local_astrocyte_depolarization): configuration,
pulse schedule and source region -- the host logic of SURVEY section 8 f3."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "examples", "local_astrocyte_depolarization", "run_stim_duration.py")


@pytest.fixture(scope="module")
def drv():
    spec = importlib.util.spec_from_file_location("run_stim_duration", DRIVER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name,period,end_time,f_value", [("baseline", 10, 100, 97), ("100-hz", 10, 100, 60),
                                                          ("300-hz", 3.3, 35, 60)])
def test_config_files_load(drv, name, period, end_time, f_value):
    cfg = drv.load_config(name)
    assert cfg["period"] == period and cfg["end_time"] == end_time and cfg["f_value"] == f_value
    assert cfg["mesh"]["cell_type"] == "tetrahedron" and cfg["save_frequency"] > 0
    assert set(drv.DEFAULTS) <= set(cfg)


def test_config_rejects_bad_schedule(drv, tmp_path):
    bad = tmp_path / "bad.yml"
    bad.write_text("period: 0\n")
    with pytest.raises(ValueError):
        drv.load_config(str(bad))


def test_pulse_schedule(drv):
    """on for pulse_width every period after delay until end_time (run_stim_duration.py:318-331,485)."""
    cfg = drv.load_config("baseline")
    on = [t for t in np.arange(0, 130, 0.1).round(1) if drv.source_is_active(float(t), cfg)]
    assert on[0] == 1.0 and 1.9 in on and 2.0 not in on and 11.5 in on and 0.9 not in on
    assert max(on) <= 100.0 and 91.0 in on and 101.0 not in on
    assert len(on) == 10 * 10      # ten pulses of ten steps


def test_three_subdomain_mesh_and_region(drv):
    cfg = drv.load_config("baseline")
    mesh, ct, ft = drv.read_mesh(cfg)
    assert set(np.unique(ct.dense())) == {0, 1, 2}
    assert {1, 2} <= set(np.unique(ft.values))
    assert abs(mesh.x[:, 0].max() - 32e-4) < 1e-12        # centimetres
    inside = drv.source_region(mesh.x, cfg)
    assert inside.any() and not inside.all()
    assert mesh.x[inside, 0].min() > cfg["x_L"] and mesh.x[inside, 0].max() < cfg["x_U"]
