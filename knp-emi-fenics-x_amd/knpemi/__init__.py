"""knpemi -- MI355X-native hot path of the KNP-EMI solver (drop-in for `src/knpemi`).

Exports the names the reference's package imports (`src/knpemi/__init__.py:1-16`);
its `__all__` lists names that do not exist and is not reproduced.
"""
from knpemi.odeSolver import MembraneModel

from knpemi.emiWeakForm import emi_system
from knpemi.emiWeakForm import create_functions_emi

from knpemi.knpWeakForm import knp_system
from knpemi.knpWeakForm import create_functions_knp

from knpemi.utils import set_initial_conditions
from knpemi.utils import setup_membrane_model
from knpemi.utils import interpolate_to_membrane
from knpemi.utils import update_ode_variables
from knpemi.utils import update_pde_variables

from knpemi.pdeSolver import create_solver_emi
from knpemi.pdeSolver import create_solver_knp

__all__ = [
    "MembraneModel", "emi_system", "create_functions_emi", "knp_system", "create_functions_knp",
    "set_initial_conditions", "setup_membrane_model", "interpolate_to_membrane",
    "update_ode_variables", "update_pde_variables", "create_solver_emi", "create_solver_knp",
]
