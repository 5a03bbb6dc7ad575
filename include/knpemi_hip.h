/* knpemi_hip.h -- C ABI of libknpemi_hip.so, the MI355X (gfx950) implementation of the
 * per-time-step hot path of adajel/knp-emi-fenics-x.
 *
 * The reference has no FFI of its own: its hot path is reached through Python calls into
 * DOLFINx/PETSc/numbalsoda.  Each entry point below names the reference call it replaces
 * (paths relative to the reference repository root).  Conventions:
 *   - every function returns 0 on success and a negative KNPEMI_E* code on failure; nothing
 *     throws across the boundary; knpemi_last_error() returns the message of the last failure
 *     on the calling thread;
 *   - the caller owns every host buffer; the library owns all device memory behind the
 *     opaque handle and frees it in knpemi_destroy();
 *   - a handle is not thread-safe; handles on different devices are independent;
 *   - kernels are enqueued on the handle's HIP stream; getters that copy to the host and
 *     knpemi_sync() block until that stream is idle.
 *   - all floating point data is IEEE binary64, all indices are int32.
 */
#ifndef KNPEMI_HIP_H
#define KNPEMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNPEMI_OK 0
#define KNPEMI_EINVAL (-1)   /* bad argument / unsupported configuration */
#define KNPEMI_EHIP (-2)     /* HIP runtime error (no device, launch failure, ...) */
#define KNPEMI_ENOMEM (-3)
#define KNPEMI_ESOLVE (-5)   /* Krylov solver did not converge (ksp_error_if_not_converged, pdeSolver.py:20,27) */
#define KNPEMI_EODE (-4)     /* LSODA reported failure on at least one membrane dof
                                (`assert success`, src/knpemi/odeSolver.py:121) */

/* cell kinds (CG-1 on each): triangle / tetrahedron = P1, hexahedron = Q1
 * (src/knpemi/emiWeakForm.py:66, examples/idealized_geometries/make_mesh_2D.py:53-55,
 *  make_mesh_3D.py:100-102) */
#define KNPEMI_TRIANGLE 0
#define KNPEMI_TETRAHEDRON 1
#define KNPEMI_HEXAHEDRON 2

/* membrane models shipped as device code (the reference's numba cfunc plug-ins):
 *   HH_SI  examples/idealized_geometries/mm_hh.py:139-227          (4 states, 22 parameters)
 *   HH_MV  examples/local_astrocyte_depolarization/mm_hh.py:130-201 (4 states, 22 parameters)
 *   GLIAL  examples/local_astrocyte_depolarization/mm_glial.py:133-205 (1 state, 23 parameters) */
#define KNPEMI_MODEL_HH_SI 0
#define KNPEMI_MODEL_HH_MV 1
#define KNPEMI_MODEL_GLIAL 2

#define KNPEMI_MAX_IONS 4   /* ionic species K = 2..4, the last one eliminated (knpWeakForm.py:53,92,131) */
#define KNPEMI_MAX_SUB 8
#define KNPEMI_MAX_MODELS 4   /* membrane models per cellular sub-domain */

/* Fields addressable with knpemi_set_field / knpemi_get_field.  `sub` is the sub-domain index
 * (0 = ECS), `idx` the ion index k or, for I_CH, model*KNPEMI_MAX_IONS + k.  Lengths are the
 * number of vertices of the sub-mesh (bulk fields) or of membrane dofs of Q_sub (PHI_M, I_CH):
 * exactly `Function.x.array` of the reference objects named on the right. */
#define KNPEMI_F_PHI 0      /* phi[tag]                      emiWeakForm.py:68   */
#define KNPEMI_F_C 1        /* c[tag][k], k < K-1            knpWeakForm.py:65   */
#define KNPEMI_F_C_PREV 2   /* c_prev[tag][k], k < K-1       knpWeakForm.py:67   */
#define KNPEMI_F_C_ELIM 3   /* ion_list[-1]['c_tag']         knpWeakForm.py:77   */
#define KNPEMI_F_PHI_M 4    /* phi_M_prev[tag]               emiWeakForm.py:78   */
#define KNPEMI_F_I_CH 5     /* mem_model['I_ch_k'][ion]      utils.py:137-141    */
#define KNPEMI_F_SOURCE 6   /* ion['f_source'] (ECS only)    knpWeakForm.py:164-166 */

/* matrices / vectors */
#define KNPEMI_A_EMI 0      /* a  of emi_system   emiWeakForm.py:138-167 */
#define KNPEMI_P_EMI 1      /* p  of emi_system   emiWeakForm.py:169-198 */
#define KNPEMI_A_KNP 2      /* a (= p) of knp_system  knpWeakForm.py:123-143,319 */
#define KNPEMI_B_EMI 0      /* L of emi_system    emiWeakForm.py:201-241 */
#define KNPEMI_B_KNP 1      /* L of knp_system    knpWeakForm.py:146-216 */

/* assemble flags */
#define KNPEMI_WANT_P 1       /* also fill P_EMI (fused with A_EMI, one pass) */
#define KNPEMI_NO_SPLITTING 2 /* splitting_scheme=False (emiWeakForm.py:234-236, knpWeakForm.py:201-206) */
/* Overlap of the ODE sweep with the EMI matrix assembly (they are independent: only the membrane
 * Robin term of b_emi needs the ODE output phi_M_prev / I_ch).
 *   KNPEMI_SKIP_MEMBRANE_RHS: knpemi_assemble_emi leaves that term out; add it later with
 *                             knpemi_assemble_emi_membrane_rhs (same flags for splitting);
 *   KNPEMI_ON_AUX_STREAM:     knpemi_assemble_emi runs on the handle's auxiliary stream, ordered after
 *                             everything enqueued so far; knpemi_join() orders the main stream after it. */
#define KNPEMI_SKIP_MEMBRANE_RHS 4
#define KNPEMI_ON_AUX_STREAM 8

typedef struct knpemi_handle knpemi_handle;

/* Flattened topology of the mixed-dimensional problem.  Replaces what the reference builds with
 * scifem.extract_submesh (run_3D.py:156-158), dolfinx function spaces (emiWeakForm.py:63-79) and
 * scifem.compute_interface_data (emiWeakForm.py:39-42).  All per-sub arrays have n_sub entries;
 * entry 0 is the ECS.  Vertices on a membrane exist once in the ECS sub-mesh and once in the
 * cell's sub-mesh.  Facet vertex a of facet f is the same physical point in facet_e, facet_i
 * and facet_q ("+" = ECS side, "-" = cell side; emiWeakForm.py:25-26). */
typedef struct {
  int32_t gdim;                      /* 2 or 3 */
  int32_t cell_kind;                 /* KNPEMI_TRIANGLE | TETRAHEDRON | HEXAHEDRON */
  int32_t n_sub;                     /* sub-domains, ECS first */
  int32_t n_ions;                    /* K = 2..4; the last ion is eliminated (run_3D.py:255-256) */
  const int32_t* n_vert;             /* [n_sub] vertices (owned + ghost) of each sub-mesh */
  const int32_t* n_cell;             /* [n_sub] */
  const double* const* x;            /* [n_sub] -> n_vert*gdim, row-major */
  const int32_t* const* cells;       /* [n_sub] -> n_cell*nv sub-mesh vertex ids */
  const int32_t* n_q;                /* [n_sub] membrane dofs of Q_sub (0 for the ECS) */
  const int32_t* n_facet;            /* [n_sub] membrane facets of the cell (0 for the ECS) */
  const int32_t* const* facet_e;     /* [n_sub] -> n_facet*nf ECS sub-mesh vertex ids */
  const int32_t* const* facet_i;     /* [n_sub] -> n_facet*nf cell sub-mesh vertex ids */
  const int32_t* const* facet_q;     /* [n_sub] -> n_facet*nf dofs of Q_sub */
  const int32_t* const* facet_model; /* [n_sub] -> n_facet index of the membrane model whose
                                        tag the facet carries (mm['ode'].tag), -1 = none */
  const int32_t* const* q_to_e;      /* [n_sub] -> n_q ECS vertex of each Q dof (utils.py:150-207) */
  const int32_t* const* q_to_i;      /* [n_sub] -> n_q cell vertex of each Q dof */
  const int32_t* n_models;           /* [n_sub] membrane models of the cell (len(mem_models)) */
  /* Optional.  Hexahedra: the three edge vectors e_t = x[1 << t] - x[0] (row t) shared by EVERY cell of a uniform box mesh
   * as its generator knows them (make_mesh_3D.py:100-102: dolfinx.mesh.create_box on a uniform grid), all zero = unknown.
   * When every cell matches it (or, unknown, the first cell) to 1e-9 the row kernels use it as the geometry of every cell
   * and stage no coordinates; a generator that supplies it gives every rank of a partitioned run the same bits.
   * Tetrahedra: the edge vectors of the uniform GRID the mesh was split from (BASELINE configs 2, 3, 5: six Kuhn tetrahedra
   * per grid cell).  When the vertices of every cell are corners of one grid cell (lattice coordinates within 1e-6 of
   * integers) and no Laplacian row is longer than 31, the row kernels take gradient products and volumes from a table of
   * the (at most eight) cell shapes and stage no coordinates; there is no fall-back to the first cell: all zero, or a cell
   * that does not fit, keeps the general kernels. */
  double uniform_cell[9];
} knpemi_problem_desc;

/* Physical parameters: the `physical_params` / `ion_list` dictionaries (run_3D.py:180-256). */
typedef struct {
  double dt, F, psi, C_M;
  double z[KNPEMI_MAX_IONS];                            /* ion['z'] */
  double D[KNPEMI_MAX_SUB][KNPEMI_MAX_IONS];            /* ion['D'][tag] */
  double rho_z;                                         /* rho['z'] (utils.py:249) */
  double rho[KNPEMI_MAX_SUB];                           /* rho[tag] */
  double C_phi;                                         /* physical_parameters['C_phi'], its own entry in the reference
                                                           (run_2D.py:187,208; emiWeakForm.py:164,231-236): the membrane
                                                           coupling and Robin datum of the EMI forms.  <= 0: C_M / dt */
} knpemi_params;

const char* knpemi_last_error(void);
int knpemi_device_count(void);

/* Build the device problem: uploads the topology, derives vertex->cell adjacency, CSR patterns
 * and row-relative scatter slots.  Replaces LinearProblem construction (pdeSolver.py:46-66,
 * 121-139: sparsity pattern + FFCx JIT). */
int knpemi_create(const knpemi_problem_desc* desc, int device, knpemi_handle** out);
void knpemi_destroy(knpemi_handle* h);
int knpemi_set_params(knpemi_handle* h, const knpemi_params* p);
int knpemi_sync(knpemi_handle* h);

/* Function I/O (`Function.x.array[:] = ...` / reading it back). */
int knpemi_set_field(knpemi_handle* h, int field, int sub, int idx, const double* host, size_t n);
int knpemi_get_field(knpemi_handle* h, int field, int sub, int idx, double* host, size_t n);

/* Assembly: the matrix/vector assembly inside problem_emi.solve() / problem_knp.solve()
 * (run_3D.py:355-356 -> dolfinx assemble_matrix/assemble_vector over FFCx kernels). */
int knpemi_assemble_emi(knpemi_handle* h, int flags);
int knpemi_assemble_knp(knpemi_handle* h, int flags);
int knpemi_assemble_emi_membrane_rhs(knpemi_handle* h, int flags);
/* The membrane-facet integrals of b_knp (knpWeakForm.py:168-214) in two parts.  Everything of the integrand that does
 * not depend on the potential -- the rational factors alpha_k, C_k g_k with phi_M and I_ch from the ODE step -- is
 * integrated by knpemi_assemble_knp_membrane_early(flags [| KNPEMI_ON_AUX_STREAM]) as soon as the ODE sweep has
 * finished, i.e. BESIDE the EMI solve; knpemi_assemble_knp(flags | KNPEMI_MEMBRANE_EARLY) then only applies the small
 * facet matrices to the jump of the potential just solved for.  Same integrals as the one-part form (1e-15 apart:
 * another summation order), one kernel less between the two solves.  Measured on MI355X it does not pay at the
 * BASELINE sizes: the membrane rows of the KNP row kernel get longer (facet -> vertex -> record loads per entry) by more
 * than the facet kernel's 13-19 us (config 2: 0.187 -> 0.193 ms per step; 995 k tets: 0.262 -> 0.275), so the
 * stepper keeps the one-part form by default. */
#define KNPEMI_MEMBRANE_EARLY 16
int knpemi_assemble_knp_membrane_early(knpemi_handle* h, int flags);
int knpemi_join(knpemi_handle* h);

/* CSR access.  A_EMI/P_EMI: square, unknown order [phi_0, phi_1, ...] (pdeSolver.py:42).
 * A_KNP: block diagonal, unknown order [c[0][0], c[0][1], c[1][0], ...] (pdeSolver.py:117). */
int knpemi_csr_dims(knpemi_handle* h, int which, int64_t* n_rows, int64_t* nnz);
int knpemi_get_csr_pattern(knpemi_handle* h, int which, int32_t* rowptr, int32_t* colind);
int knpemi_get_csr_values(knpemi_handle* h, int which, double* vals);
int knpemi_get_rhs(knpemi_handle* h, int which, double* b);
/* Caller-supplied values of an operator / right-hand side, in the layout of knpemi_get_csr_values / knpemi_get_rhs
 * (petsc4py Mat.setValuesCSR / Vec.setArray on `problem.A` / `problem.b`, pdeSolver.py:56-66 expose both objects).
 * Synchronous. */
int knpemi_set_csr_values(knpemi_handle* h, int which, const double* vals);
int knpemi_set_rhs(knpemi_handle* h, int which, const double* b);
/* Device-resident views for on-GPU consumers (solvers, RCCL halo exchange through torch). */
int knpemi_device_csr(knpemi_handle* h, int which, const int32_t** rowptr, const int32_t** colind,
                      const double** vals);
int knpemi_device_rhs(knpemi_handle* h, int which, const double** b);
/* Write a solver result back into the bulk fields: x has the unknown order of the system
 * (`which` = KNPEMI_B_EMI -> phi, KNPEMI_B_KNP -> c).  on_device != 0: x is a device pointer (one launch, nothing
 * synchronised).  With KNPEMI_OPT_FUSE_UPDATE the KNP write-back is followed by (device: fused with) the end-of-step
 * update knpemi_update_pde, on both paths. */
int knpemi_set_solution(knpemi_handle* h, int which, const double* x, int on_device);
int knpemi_get_solution(knpemi_handle* h, int which, double* x);

/* Linear solves on the device (adjacent to the hot path): the KSP solve inside
 * problem_emi.solve() / problem_knp.solve() with the iterative options of pdeSolver.py:24-35,99-110.
 * EMI: preconditioned CG on A_EMI x = B_EMI with the constant null space projected out (:74-78);
 * KNP: right-preconditioned BiCGStab on the block-diagonal system.  Both start from the current fields
 * (ksp_initial_guess_nonzero), stop at ||r|| <= max(atol, rtol ||b||) and write the solution back into
 * phi / c.  Returns KNPEMI_ESOLVE when maxit is reached. */
enum { KNPEMI_PC_JACOBI = 0, KNPEMI_PC_AMG = 1 };
int knpemi_solve_emi(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres);
int knpemi_solve_knp(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres);
/* Replace the initial guess of the next solve of `which` (the current phi / c, i.e. the previous solution:
 * ksp_initial_guess_nonzero, pdeSolver.py:26,101) by the extrapolation of the last solutions: 3 x_n - 3 x_(n-1) + x_(n-2)
 * once three are known, 2 x_n - x_(n-1) after two; the first call only records x_n.  Call once per time step, before
 * the solve. */
int knpemi_extrapolate_guess(knpemi_handle* h, int which);
/* Preconditioner of the device solve of system `which` (KNPEMI_B_EMI / KNPEMI_B_KNP): the counterpart of
 * pc_type / pc_hypre_type in pdeSolver.py:27-34,102-109.  KNPEMI_PC_AMG: smoothed-aggregation V(1,1) cycle,
 * strength threshold `theta` (<= 0: default 0.08); the hierarchy is (re)built at the next solve and kept for
 * the following time steps (rebuilt when the iteration count has doubled).  Default: AMG for both. */
int knpemi_solver_setup(knpemi_handle* h, int which, int precond, double theta);
/* levels / operator complexity / number of builds of the AMG hierarchy of `which` (0 levels: not built). */
int knpemi_solver_info(knpemi_handle* h, int which, int* levels, double* op_complexity, int* builds);

/* Membrane ODEs: MembraneModel (src/knpemi/odeSolver.py:6-188).  `states`/`params` are the
 * row-major [n_q][n_states|n_params] tables `MembraneModel.states/.parameters`. */
int knpemi_ode_bind(knpemi_handle* h, int sub, int model, int model_id, int n_states, int n_params);
/* A membrane model that brings its own right-hand side (the reference accepts any module with a numba cfunc
 * `rhs_numba(t, states, values, parameters)`, odeSolver.py:96, e.g. examples/benchmark/mm_glial.py:127-215).
 * `rhs_source` is HIP source that defines
 *     __device__ void rhs(double t, const double* states, double* values, double* parameters);
 * with the cfunc's semantics (parameters is the in/out row of the dof: what the last call stores in I_ch_* is handed
 * to the PDEs).  It is compiled for gfx950 with hipRTC under the sweep kernel of the shipped models; one lane per
 * state component for 1, 2, 4 or 8 states, one thread per dof otherwise.  knpemi_ode_compile_source only compiles
 * (no device needed) and returns the compiler's messages. */
int knpemi_ode_bind_source(knpemi_handle* h, int sub, int model, int n_states, int n_params, const char* rhs_source);
int knpemi_ode_compile_source(int n_states, int n_params, const char* rhs_source, char* log, size_t log_len);
int knpemi_ode_set_tables(knpemi_handle* h, int sub, int model, const double* states,
                          const double* params);
int knpemi_ode_get_tables(knpemi_handle* h, int sub, int model, double* states, double* params);
/* stimulus_mask[n_q] (may be NULL = everywhere) and (param index, value) pairs written into the
 * masked rows before every step (odeSolver.py:98-112). */
int knpemi_ode_set_stimulus(knpemi_handle* h, int sub, int model, const uint8_t* stimulus_mask,
                            int n_pairs, const int32_t* param_idx, const double* values);
/* One fused launch of update_ode_variables (utils.py:210-235: with KNPEMI_ODE_SET_TRACES the
 * concentration traces go into the "<ion>_e"/"<ion>_i" parameter columns, with KNPEMI_ODE_SET_V
 * V <- phi_M_prev, i.e. the k > 0 branch of utils.py:233), MembraneModel.step_lsoda
 * (odeSolver.py:92-127) and the copy-back of run_3D.py:104-109 (phi_M_prev <- V, I_ch_k <-
 * currents).  ion_param[3*K] = parameter indices of "<ion>_e", "<ion>_i", "I_ch_<ion>" for each
 * ion (ode.parameter_indices), v_index = ode.state_indices('V'). */
#define KNPEMI_ODE_SET_V 1
#define KNPEMI_ODE_SET_TRACES 2
/* run the sweep on the auxiliary stream (fork after what is queued on the main stream; knpemi_join waits for it):
 * lets a caller keep the LONGER of {ODE sweep, EMI matrix assembly} on the main stream, so that the work that follows
 * the join does not pay the cross-stream signal latency */
#define KNPEMI_ODE_ON_AUX_STREAM 4
/* the same on a second auxiliary stream: the sweeps of different membrane models (neuron, glia, ...) are independent
 * of each other and each fills a fraction of the chip only, so they run side by side */
#define KNPEMI_ODE_ON_AUX2_STREAM 8
int knpemi_ode_step(knpemi_handle* h, int sub, int model, double t0, double dt, double rtol,
                    double atol, int flags, const int32_t* ion_param, int v_index);
/* RHS evaluations / internal steps / failed dofs summed over all knpemi_ode_step launches since
 * the previous call (the counters are reset by this call). */
int knpemi_ode_stats(knpemi_handle* h, int sub, int model, int64_t* n_rhs, int64_t* n_steps,
                     int32_t* n_failed);

/* Diagnostics (no reference counterpart): with KNPEMI_ODE_STAMPS=1 in the environment the sweep runs a stamped build of
 * the kernel; per workgroup 12 cycle sums (loop head, TOP, PRED, RHS, CORR, ERR up to the order selection, prologue, -,
 * order selection, new coefficients + rescaling, rest of ERR, -) and 12 counts.  Returns
 * the number of workgroups copied (<= max_blocks). */
int knpemi_debug_ode_stamps(knpemi_handle* h, int sub, int model, uint64_t* out, int max_blocks);

/* Diagnostics (no reference counterpart): evaluates one of the ODE sweep's device math helpers (csrc/lsoda_core.h) over
 * host arrays on the current device, so that tests can bound their error against the C library: op 0: kn_div(a, b),
 * 1: kn_exp(a), 2: kn_powr(a, b), 3: kn_log(a).  a, b, out: n doubles each (b is ignored by ops 1 and 3). */
int knpemi_debug_math(int op, int n, const double* a, const double* b, double* out);

/* Diagnostics (no reference counterpart): `links` dependent trivial kernels on the handle's stream, `reps` times, launched
 * one by one or replayed from a captured hipGraph; *us_per_kernel = host wall time per kernel.  kind 0: an empty one-wave
 * kernel, 1: y = x + 1 over n doubles, 2: the one-block start kernel of the fused Krylov loops.  What a dependent launch
 * costs inside THIS process and on THIS stream, next to tools/probes/kernel_chain.hip which measures a bare process. */
int knpemi_debug_launch_chain(knpemi_handle* h, int kind, int n, int links, int reps, int use_graph, double* us_per_kernel);

/* Diagnostics (no reference counterpart): which geometry specialisations knpemi_create selected for the row kernels.
 * *flags: bit 0 = lattice tetrahedra of a uniform grid (shape table, no coordinates staged), bit 1 = hexahedra that are all
 * parallelepipeds, bit 2 = ... and all the same box (uniform hexahedral kernels). */
int knpemi_debug_geometry(knpemi_handle* h, int* flags);

/* End-of-step update: update_pde_variables (utils.py:238-295): c_prev <- c, eliminated ion from
 * electroneutrality, phi_M_prev <- tr(phi_i) - tr(phi_e). */
int knpemi_update_pde(knpemi_handle* h);

/* Options of a handle (device-resident loops).
 * KNPEMI_OPT_FUSE_UPDATE (0/1): update_pde_variables follows problem_knp.solve() directly in the reference's loop
 *   (run_3D.py:356,362); with this option the write-back kernel of knpemi_solve_knp -- and of
 *   knpemi_set_solution(KNPEMI_B_KNP, device pointer) -- also performs that update (same arithmetic, one launch
 *   fewer per step), and the caller does not call knpemi_update_pde.
 * KNPEMI_OPT_FUSE_MEMBRANE (0/1, default 0): 1 evaluates the membrane-facet integrals of b_knp
 *   (knpWeakForm.py:178-214) inside the KNP row kernel -- each row block integrates the (facet, local vertex) entries
 *   of its rows before it sums them, same arithmetic and order as the stand-alone facet kernel, bit-identical b_knp --
 *   instead of in a launch of their own that hands them over through HBM.  Measured on MI355X it does not pay: the
 *   degree-6 facet quadrature lands on the few row blocks that own membrane rows, so the row kernel's tail grows by as
 *   much as (24,794 dofs: 17.0 + 16.3 us apart, 34.2 us fused) or more than (219,542 dofs: 59.3 + 18.7 us apart,
 *   107 us fused) the stand-alone kernel costs spread over all CUs.  Kept as an option for small, launch-bound cases.
 *   It needs row blocks of consecutive rows (the membrane entries of a block are then one range): create the handle
 *   with KNPEMI_BLOCK_CLASSIC=1 in the environment, otherwise the option is refused with KNPEMI_EINVAL. */
#define KNPEMI_OPT_FUSE_UPDATE 1
#define KNPEMI_OPT_FUSE_MEMBRANE 2
/* KNPEMI_OPT_PROFILE_STRIDE (n >= 1, default 1): knpemi_profile brackets every n-th launch of a selected kernel only. */
#define KNPEMI_OPT_PROFILE_STRIDE 3
/* KNPEMI_OPT_KNP_MIN_IT (n >= 0, default 0): knpemi_solve_knp performs at least n iterations before a residual below
 * the target ends it (`ksp_min_it`: 5 in the reference's iterative options of the concentration solve, pdeSolver.py:101;
 * knpemi.pdeSolver.create_solver_knp sets it).  A residual that has vanished exactly still ends the solve. */
#define KNPEMI_OPT_KNP_MIN_IT 4
/* KNPEMI_OPT_FOLD_MEMBRANE (0/1, default 1): the launch that writes the potential back at the end of knpemi_solve_emi
 * (single rank, fused loop) or of knpemi_set_solution(KNPEMI_B_EMI, on_device) also forms the membrane-facet integrals of
 * b_knp (knpWeakForm.py:168-214) for that potential, so knpemi_assemble_knp launches the row kernel only: one dependent
 * launch fewer between the two solves.  The integrals are used only while none of their inputs has changed since; 0 keeps
 * the facet kernel a launch of its own (what bench.py times as the facet-assembly kernel). */
#define KNPEMI_OPT_FOLD_MEMBRANE 5
/* KNPEMI_OPT_KNP_METHOD (default 0): 0 = right-preconditioned BiCGStab, convergence on the true residual (fewest launches per
 * V-cycle); 1 = what PETSc's defaults make of the reference's `ksp_type gmres` (pdeSolver.py:100): GMRES with restart 30, left
 * preconditioning, classical Gram-Schmidt, convergence on the preconditioned residual norm relative to |M^-1 b|; the
 * iteration count and KNPEMI_OPT_KNP_MIN_IT then count GMRES iterations as `ksp_min_it` / getIterationNumber() do.
 * Single rank, AMG preconditioner (the fused loops); otherwise the option is ignored. */
#define KNPEMI_OPT_KNP_METHOD 6
/* KNPEMI_OPT_EMI_NORM (default 0): convergence test of the potential solve's CG: 0 = true residual |b - A x| <= max(atol,
 * rtol |b|); 1 = what PETSc's defaults make of the reference's `ksp_type cg` (pdeSolver.py:60-72; KSPCG: left preconditioning,
 * KSP_NORM_PRECONDITIONED): |M^-1 r| <= max(atol, rtol |M^-1 b|), the residual norm reported is that one.  Single rank,
 * AMG preconditioner (the fused loop); otherwise the option is ignored. */
#define KNPEMI_OPT_EMI_NORM 7
int knpemi_set_option(knpemi_handle* h, int option, int value);

/* Nodal trace of an (ECS, cell) pair of bulk functions onto Q_sub: interpolate_to_membrane
 * (utils.py:150-207).  u_e has n_vert[0] entries, u_i n_vert[sub]; q_e, q_i receive n_q[sub]. */
int knpemi_trace(knpemi_handle* h, int sub, const double* u_e, const double* u_i, double* q_e,
                 double* q_i);

/* Multi-GPU forward halo (owner -> ghost) of dof fields; replaces
 * Function.x.scatter_forward() (utils.py:100,199,204,254,293).  idx_dev / buf_dev are DEVICE pointers
 * (the caller moves buf between GPUs, e.g. torch.distributed send/recv over RCCL).
 * kind 0 (bulk): idx = global vertex ids (sub-mesh vertex + offset of its sub-domain), 5 doubles per
 *   entry: the K concentrations (solved ones as c_prev, the eliminated one last) and phi;
 * kind 1 (membrane): idx = global Q-dof ids, 1 + 3*n_model_slots doubles per entry: phi_M_prev and
 *   the I_ch_k (KNPEMI_MAX_IONS slots per model) of every membrane model. */
int knpemi_halo_width(knpemi_handle* h, int kind);
int knpemi_halo_pack(knpemi_handle* h, int kind, const int32_t* idx_dev, int n, double* buf_dev);
int knpemi_halo_unpack(knpemi_handle* h, int kind, const int32_t* idx_dev, int n, const double* buf_dev);

/* Distributed solves on a partitioned problem (one handle per rank; the reference runs its KSP solves on the MPI
 * communicator of the mesh, pdeSolver.py:24-35,74-78,99-110).  The library knows which local vertices this rank owns;
 * the caller supplies the two communication steps, both ordered on knpemi_stream(h):
 *   allreduce(ctx, n): sum the first n doubles of `reduce_buf_dev` over all ranks, in place;
 *   halo(ctx, vec_dev, which): forward halo (owner -> ghost) of a device vector in the unknown order of system `which`
 *     (KNPEMI_B_EMI: one value per local vertex; KNPEMI_B_KNP: the block order [sub-domain][ion][vertex]).
 * knpemi_solve_emi / knpemi_solve_knp then run the same Krylov methods on the global system: SpMV over the owned rows
 * after a halo of its argument, dot products over the owned entries summed with `allreduce`, and each rank's
 * smoothed-aggregation V-cycle on its own diagonal block as the preconditioner (block Jacobi over the ranks).
 * `owned` is a host array with one byte per local vertex (global numbering of this handle: sub-mesh vertex + offset
 * of its sub-domain), 1 = owned.  owned == NULL switches back to the single-rank solves. */
typedef int (*knpemi_allreduce_fn)(void* ctx, int n);
typedef int (*knpemi_halo_fn)(void* ctx, void* vec_dev, int which);
int knpemi_set_distributed(knpemi_handle* h, const uint8_t* owned, void* reduce_buf_dev, knpemi_allreduce_fn allreduce,
                           knpemi_halo_fn halo, void* ctx);
/* gather / scatter of entries of a device vector (the pack / unpack of the halo of a solver vector) */
/* Two-level variant of the distributed EMI preconditioner: next to every rank's AMG cycle a coarse space of one
 * piecewise-constant function per chunk of consecutive vertices of every sub-domain of every rank (as many chunks as
 * fit 64 functions in all) -- A_c = Phi^T A Phi is built with <= 64 SpMVs when the hierarchy is (re)built, every
 * application costs one all-reduce of <= 64 doubles.  Call after knpemi_set_distributed; the reduction buffer then
 * needs 8 + 64 doubles.  world <= 1 switches it off. */
int knpemi_set_distributed_coarse(knpemi_handle* h, int rank, int world);
int knpemi_vec_gather(knpemi_handle* h, const void* vec_dev, const int32_t* idx_dev, int n, void* buf_dev);
int knpemi_vec_scatter(knpemi_handle* h, void* vec_dev, const int32_t* idx_dev, int n, const void* buf_dev);

/* ---- RCCL transport inside the library (the reference: MPI inside DOLFINx / PETSc, Function.x.scatter_forward() and the
 * parallel KSP of pdeSolver.py:24-35) ----------------------------------------------------------------------------------
 * One process per GPU.  Rank 0 obtains a unique id (128 bytes), the caller distributes it over whatever rendezvous it
 * has, every rank calls knpemi_comm_init (collective).  knpemi_comm_sendrecv posts, in ONE RCCL group on the handle's
 * stream, for every part p: send_buf[send_off[p], +send_cnt[p]) -> peer[p] and recv_buf[recv_off[p], +recv_cnt[p]) <-
 * peer[p] (device buffers, counts in doubles): with knpemi_halo_pack before and knpemi_halo_unpack after it this is the
 * forward halo of a step, stream-ordered, no host synchronisation, no Python.  knpemi_comm_allreduce sums n doubles of a
 * device buffer over the ranks.  knpemi_comm_allreduce_hook / knpemi_comm_halo_hook have the signatures of
 * knpemi_allreduce_fn / knpemi_halo_fn and take the handle as ctx: passed to knpemi_set_distributed (after
 * knpemi_comm_set_vector_plan for both systems) they keep the distributed Krylov solves inside the library.
 * RCCL is resolved at run time (the copy already loaded in the process, else /opt/rocm/lib/librccl.so). */
int knpemi_comm_unique_id(char* out, size_t len);
int knpemi_comm_init(knpemi_handle* h, int rank, int world, const char* id_bytes, size_t len);
int knpemi_comm_sendrecv(knpemi_handle* h, const double* send_buf_dev, double* recv_buf_dev, int n_parts,
                         const int32_t* peer, const int64_t* send_off, const int64_t* send_cnt, const int64_t* recv_off,
                         const int64_t* recv_cnt);
int knpemi_comm_allreduce(knpemi_handle* h, double* buf_dev, int n);
int knpemi_comm_set_vector_plan(knpemi_handle* h, int which, const int32_t* send_idx_dev, int n_send,
                                const int32_t* recv_idx_dev, int n_recv, double* send_buf_dev, double* recv_buf_dev,
                                int n_parts, const int32_t* peer, const int64_t* send_off, const int64_t* send_cnt,
                                const int64_t* recv_off, const int64_t* recv_cnt);
int knpemi_comm_allreduce_hook(void* ctx, int n);
int knpemi_comm_halo_hook(void* ctx, void* vec_dev, int which);

/* Per-kernel HIP-event profiling on the handle's stream: every launch of a kernel whose bit is set
 * in `kernel_mask` is bracketed by an event pair; knpemi_profile_read() synchronises, returns the
 * number of bracketed launches and their summed duration, and resets the accumulator. */
#define KNPEMI_K_ODE 0           /* ode_step_kernel      */
#define KNPEMI_K_EMI_ROWS 1      /* emi_rows_kernel      */
#define KNPEMI_K_KNP_ROWS 2      /* knp_rows_kernel      */
#define KNPEMI_K_KNP_MEMBRANE 3  /* knp_membrane_kernel  */
#define KNPEMI_K_UPDATE 4        /* update_pde_kernel    */
#define KNPEMI_K_EMI_MEMBRANE 5  /* emi_membrane_rhs_kernel */
#define KNPEMI_N_KERNELS 6
int knpemi_profile(knpemi_handle* h, uint32_t kernel_mask);
int knpemi_profile_read(knpemi_handle* h, int kernel, int64_t* launches, double* total_ms);

/* Stream-event timing of a region on the handle's stream (bench.py / rocprof cross-check). */
int knpemi_timer_start(knpemi_handle* h);
int knpemi_timer_stop_ms(knpemi_handle* h, double* ms);
/* HIP stream (hipStream_t) the handle enqueues on, for external event timing / ordering. */
void* knpemi_stream(knpemi_handle* h);

/* ---- DG(P1) + symmetric interior penalty variant -------------------------------------
 * The reference's README (README.md:5-7) describes a "DG fem method" and its 2D mesh script keeps the "interior
 * facets are tagged 0" convention of that method (examples/idealized_geometries/make_mesh_2D.py:88-90), but the
 * code under src/knpemi is continuous Galerkin on sub-meshes: there is NO reference interface these entry points
 * replace.  They provide the same step on ONE mesh with broken P1 functions: the volume terms of
 * emiWeakForm.py:138-241 / knpWeakForm.py:123-166 cell by cell, interior-penalty and upwind terms on the interior
 * facets of every sub-domain, and the reference's membrane terms (emiWeakForm.py:160-165,228-239,
 * knpWeakForm.py:168-214) on the tagged facets, with phi_M and I_ch living at the vertices of each membrane facet.
 * Triangles and tetrahedra (broken P1) and hexahedra (broken Q1: the cell type of the reference's own 3-D idealized
 * mesh, make_mesh_3D.py:100-102; volume terms by the 2 x 2 x 2 Gauss rule, facet terms by the 2 x 2 rule with every
 * quantity taken at the point, membrane terms by the reference's quadrilateral rules).  Dof (cell c, local vertex j)
 * = c * nv + j; membrane node (facet f, vertex a) = f * nf + a; CSR rows hold one nv-wide block per cell (the cell
 * itself and its facet neighbours, in increasing cell order).  The K - 1 concentration systems share the pattern of
 * the potential system. */
typedef struct knpemi_dg knpemi_dg;

typedef struct {
  int32_t cell_kind;            /* KNPEMI_TRIANGLE | KNPEMI_TETRAHEDRON | KNPEMI_HEXAHEDRON (broken Q1; local vertex j at the
                                   reference point whose coordinate along axis a is bit a of j) */
  int32_t n_sub;                /* sub-domains, ECS = 0 */
  int32_t n_ions;               /* K = 2..4, the last one eliminated */
  int64_t n_cells, n_vertices, n_mem_facets;
  const double* x;              /* [n_vertices][gdim] */
  const int32_t* cells;         /* [n_cells][nv] vertex ids */
  const int32_t* cell_sub;      /* [n_cells] sub-domain index of every cell */
  const int32_t* mem_facets;    /* [n_mem_facets][nf] vertex ids of the membrane facets: every one separates an ECS
                                   cell from a cell of a sub-domain > 0; every other facet between two cells must
                                   lie inside one sub-domain */
} knpemi_dg_desc;

typedef struct {
  double dt, F, psi, C_M;
  double gamma;                                         /* interior penalty parameter (10 is a safe default) */
  double z[KNPEMI_MAX_IONS];
  double D[KNPEMI_MAX_SUB][KNPEMI_MAX_IONS];
  double rho_z;
  double rho[KNPEMI_MAX_SUB];
} knpemi_dg_params;

#define KNPEMI_DG_C 0         /* previous-step concentration of ion idx < K (the eliminated ion last)  [n_dofs] */
#define KNPEMI_DG_PHI 1       /* potential                                                            [n_dofs] */
#define KNPEMI_DG_PHI_M 2     /* phi_M_prev at the membrane nodes                                     [n_mem_nodes] */
#define KNPEMI_DG_I_CH 3      /* channel current of ion idx < K at the membrane nodes                 [n_mem_nodes] */
#define KNPEMI_DG_SOURCE 4    /* source term of solved ion idx < K - 1 (used in ECS cells only)       [n_dofs] */

int knpemi_dg_create(const knpemi_dg_desc* desc, int device, knpemi_dg** out);
void knpemi_dg_destroy(knpemi_dg* h);
int knpemi_dg_set_params(knpemi_dg* h, const knpemi_dg_params* p);
int knpemi_dg_dims(knpemi_dg* h, int64_t* n_dofs, int64_t* nnz, int64_t* n_mem_nodes);
int knpemi_dg_get_pattern(knpemi_dg* h, int32_t* rowptr, int32_t* colind);
/* dofs of the ECS cell / of the intracellular cell that sit at every membrane node */
int knpemi_dg_get_membrane_dofs(knpemi_dg* h, int32_t* dof_e, int32_t* dof_i);
int knpemi_dg_set_field(knpemi_dg* h, int field, int idx, const double* host, size_t n);
int knpemi_dg_get_field(knpemi_dg* h, int field, int idx, double* host, size_t n);
/* flags: KNPEMI_NO_SPLITTING.  assemble_emi fills A_emi, b_emi from the concentrations, phi_M (and I_ch without the
 * splitting scheme); assemble_knp fills the K - 1 matrices and right-hand sides from the concentrations, the potential,
 * phi_M and I_ch.  One kernel launch each. */
int knpemi_dg_assemble_emi(knpemi_dg* h, int flags);
int knpemi_dg_assemble_knp(knpemi_dg* h, int flags);
/* which: 0 = potential system, 1 + k = concentration system of solved ion k */
int knpemi_dg_get_values(knpemi_dg* h, int which, double* vals);
int knpemi_dg_get_rhs(knpemi_dg* h, int which, double* b);
int knpemi_dg_device_system(knpemi_dg* h, int which, const int32_t** rowptr, const int32_t** colind, const double** vals,
                            const double** b);
/* End of step (utils.py:238-295): c_prev <- c_new ([K-1][n_dofs], host or device pointer), eliminated ion from
 * electroneutrality dof by dof, phi_M <- phi_i - phi_e at the membrane nodes. */
int knpemi_dg_update(knpemi_dg* h, const double* c_new, int on_device);
/* Device solves of the DG systems (the KSP solves of pdeSolver.py:24-35,74-78,99-110 with the iterative options,
 * SURVEY section 8 f1 for the f4 variant): CG on the potential system with the constants projected out, BiCGStab on the
 * K - 1 concentration systems, both preconditioned by the library's smoothed-aggregation AMG whose first coarse level is
 * the continuous P1 space of every sub-domain (auxiliary space) under a damped-Jacobi smoother on the broken dofs.
 * Start from the current potential / the previous concentrations, stop at ||r|| <= max(atol, rtol ||b||),
 * KNPEMI_ESOLVE at maxit.  solve_emi writes the potential field; solve_knp keeps the solution on the device and, with
 * update != 0, runs knpemi_dg_update on it; knpemi_dg_get_solution copies it out ([K-1][n_dofs]).  Single rank. */
int knpemi_dg_solve_emi(knpemi_dg* h, double rtol, double atol, int maxit, int* iters, double* relres);
/* On a cell partition (knpemi.dg.DGSlab; the reference's parallel KSP, SURVEY section 8 e + f1, for the f4 variant): the two
 * solves become solves of the global systems exactly as knpemi_set_distributed arranges for the CG path -- owned rows
 * only, halo of every SpMV argument, all-reduced dot products, per-rank auxiliary-space AMG.  `owned`: one byte per local
 * broken dof, 1 = dof of an owned cell (NULL: back to single-rank solves); vector orders for `halo`: KNPEMI_B_EMI one value
 * per dof, KNPEMI_B_KNP [solved ion][dof].  knpemi_dg_solver_handle returns the knpemi_handle the solves run on, for
 * knpemi_vec_gather / knpemi_vec_scatter in the halo hook (NULL on error). */
int knpemi_dg_set_distributed(knpemi_dg* h, const uint8_t* owned, void* reduce_buf_dev, knpemi_allreduce_fn allreduce,
                              knpemi_halo_fn halo, void* ctx);
void* knpemi_dg_solver_handle(knpemi_dg* h);
int knpemi_dg_solve_knp(knpemi_dg* h, double rtol, double atol, int maxit, int* iters, double* relres, int update);
int knpemi_dg_get_solution(knpemi_dg* h, double* c_host);
/* on != 0: both solves start from 2 x_n - x_(n-1) (the last two solutions) instead of x_n, as knpemi_extrapolate_guess
 * does for the CG path; the stopping criterion is unchanged. */
int knpemi_dg_set_extrapolation(knpemi_dg* h, int on);
/* Membrane ODE sweep over the membrane nodes with one of the built-in models (KNPEMI_MODEL_*): same kernel, tables and
 * flags as knpemi_ode_step; states / params are [n_mem_nodes][n_states | n_params] row-major on the host. */
int knpemi_dg_ode_bind(knpemi_dg* h, int model_id, int n_states, int n_params, const double* states, const double* params,
                       const int32_t* ion_param, int v_index);
int knpemi_dg_ode_step(knpemi_dg* h, double t0, double dt, double rtol, double atol, int flags);
int knpemi_dg_ode_get_tables(knpemi_dg* h, double* states, double* params);
int knpemi_dg_ode_stats(knpemi_dg* h, int64_t* n_rhs, int64_t* n_steps, int64_t* n_failed);
/* Cell-partitioned runs: every rank holds its cells plus one layer of ghost cells across the cut facets.  The ghost
 * cells' dofs (five field doubles each: the K concentrations and the potential) are refreshed from their owners once
 * per step: pack the listed dofs into a device buffer [n][5] / unpack it on the receiving side (device index lists,
 * stream-ordered; the transport is the caller's: RCCL point-to-point in knpemi/dg.py, as Function.x.scatter_forward()
 * of a DG function does under MPI in DOLFINx).  Membrane nodes of ghost cells are integrated redundantly. */
int knpemi_dg_halo_pack(knpemi_dg* h, const int32_t* idx_dev, int n, double* buf_dev);
int knpemi_dg_halo_unpack(knpemi_dg* h, const int32_t* idx_dev, int n, const double* buf_dev);
/* the library's RCCL transport for that halo, on the DG handle's stream (see knpemi_comm_init / knpemi_comm_sendrecv) */
int knpemi_dg_comm_init(knpemi_dg* h, int rank, int world, const char* id_bytes, size_t len);
int knpemi_dg_comm_sendrecv(knpemi_dg* h, const double* send_buf_dev, double* recv_buf_dev, int n_parts,
                            const int32_t* peer, const int64_t* send_off, const int64_t* send_cnt, const int64_t* recv_off,
                            const int64_t* recv_cnt);
int knpemi_dg_sync(knpemi_dg* h);
/* average duration (ms) of `reps` back-to-back launches of one assembly kernel (0 = potential, 1 = concentrations),
 * measured with HIP events on the stream the kernel is launched on */
int knpemi_dg_time_kernel(knpemi_dg* h, int which, int flags, int reps, double* avg_ms);
/* HIP event brackets of the assembly kernels inside a running time loop (as knpemi_profile): on = n >= 1 brackets
 * every n-th launch of each kernel, 0 switches them off;
 * profile_read synchronises, returns the number of bracketed launches of kernel `which` and their summed duration,
 * and resets the accumulator */
int knpemi_dg_profile(knpemi_dg* h, int on);
int knpemi_dg_profile_read(knpemi_dg* h, int which, int64_t* launches, double* total_ms);
/* HIP stream the handle enqueues on */
void* knpemi_dg_stream(knpemi_dg* h);

#ifdef __cplusplus
}
#endif
#endif /* KNPEMI_HIP_H */
