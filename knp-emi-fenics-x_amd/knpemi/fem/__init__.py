"""Host-side mesh / function layer used by the knpemi MI355X hot path."""
from .mesh import (CELL_INFO, CellType, EntityMap, Mesh, MeshTags, compute_interface_data,
                   create_box, create_rectangle, create_unit_square, exterior_facet_indices,
                   extract_submesh, find_interface, locate_entities, match_facets, meshtags,
                   transfer_meshtags_to_submesh)
from .function import Constant, Function, FunctionSpace, Vector, as_float, functionspace
from .idealized import make_mesh_2D, make_mesh_3D, make_mesh_mms
from .xdmf import XDMFFile
