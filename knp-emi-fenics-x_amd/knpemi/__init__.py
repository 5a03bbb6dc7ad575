"""knpemi -- MI355X-native hot path of the KNP-EMI solver (drop-in for `src/knpemi`).

Re-exports the twelve public names of the reference package (`src/knpemi/__init__.py:1-16`).  The reference's
`__all__` names several functions that do not exist; the list below is the set that can actually be imported
from it.  The device-resident loop of this implementation lives in `knpemi.stepper`.
"""
from .emiWeakForm import create_functions_emi, emi_system
from .knpWeakForm import create_functions_knp, knp_system
from .odeSolver import MembraneModel
from .pdeSolver import create_solver_emi, create_solver_knp
from .utils import (interpolate_to_membrane, set_initial_conditions, setup_membrane_model, update_ode_variables,
                    update_pde_variables)

__all__ = sorted([
    "MembraneModel", "create_functions_emi", "create_functions_knp", "create_solver_emi", "create_solver_knp",
    "emi_system", "interpolate_to_membrane", "knp_system", "set_initial_conditions", "setup_membrane_model",
    "update_ode_variables", "update_pde_variables",
])
