// Internal data structures shared by the host-side builder/API and the gfx950 kernels.
// Not part of the C ABI (see include/knpemi_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include "block_spmv.h"
#include <stdint.h>

#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/knpemi_hip.h"

#define KN_MAXK KNPEMI_MAX_IONS
#define KN_MAXSUB KNPEMI_MAX_SUB
#define KN_REC 8          // doubles per vertex record: x y z c3 | c0 c1 c2 phi  (ion k lives in slot KN_CSLOT(k);
                          // K = 3: c2 is the eliminated ion and slot 3 is unused, as in every reference driver)
#define KN_CSLOT(k) ((k) < 3 ? 4 + (k) : 3)
#define KN_BLOCK 256          // threads per row-kernel workgroup (rows per block = KN_BLOCK / lanes-per-row)
#define KN_SLICE 64       // rows per sliced-ELL slice == wavefront width on gfx950
// consecutive rows per chunk of a row block (kernels_assemble.hip: BlkRows).  Measured at 995 k tets / 166 k hexahedra:
// 16-row chunks give the simplex kernels fewer, longer output pieces (emi_rows 46.6 -> 43.3 us, knp_rows 45.8 -> 44.7 us;
// 266 instead of 233 distinct vertices per block), the hexahedral ones lose with them (blocks closed early by the bound on
// the distinct vertices are half empty: knp_rows_hex 61 -> 71 us); 4-row chunks lose everywhere.
#define KN_CHUNK_SIMPLEX 16
#define KN_CHUNK_HEX 8

// Per-sub-domain constants folded on the host from knpemi_params (double arithmetic identical
// to what the kernels would do per cell).
struct KnSubConst {
  double kap[KN_MAXK];   // F * psi * z_k^2 * D_k^s      (emiWeakForm.py:103)
  double sig[KN_MAXK];   // F * z_k * D_k^s              (emiWeakForm.py:217)
  double D[KN_MAXK];     // D_k^s
  double zpsiD[KN_MAXK]; // z_k * psi * D_k^s            (knpWeakForm.py:141)
  double az2D[KN_MAXK];  // D_k^s z_k^2                  (knpWeakForm.py:97)
  double rho_term;       // -(1/z_K) * rho_z * rho^s     (utils.py:249)
};

struct KnConsts {
  int n_sub, K;
  int voff[KN_MAXSUB + 1];   // global vertex offset of each sub-domain
  int qoff[KN_MAXSUB + 1];   // global Q-dof offset
  KnSubConst sc[KN_MAXSUB];
  double dt, inv_dt, F, psi, C_M, C_phi;
  double z[KN_MAXK];
  double elim_coef[KN_MAXK]; // -(z_k / z_K)            (utils.py:258)
  int splitting;
};

// Device views (raw pointers; owned by knpemi_handle).
struct KnDev {
  int Ntot, nctot, NQtot, nftot;
  int nblocks;                // row blocks; a block never straddles two sub-domains
  const int* blk_rng;         // [nblocks][KN_BLOCK / lpr / chunk][6] chunks of consecutive rows (BlkRows)
  const int* blk_sub;         // [nblocks] sub-domain of the block
  const int4* blk_info;       // [nblocks][4]: {first row, rows, sub, 0} {EMI seg length, offset into ent_loc,
                              //   Laplacian seg length, offset into blk_uverts} {slice entry bases / 64}
                              //   {4 x uint8 slice steps, #distinct vertices, 0, 0}
  const int* blk_uverts;      // concatenated per-block sorted lists of the distinct vertices its rows touch
  const uint16_t* ent_loc;    // [nnzL] position of every Laplacian entry's vertex in its block's list, stored in the
                              //   order of the blocks' concatenated Laplacian segments
  const int4* row_info;       // [Ntot]: offsets in the block's concatenated segments {EMI row start, (that + lapoff[g]) |
                              //          n_membrane_entries << 16, Laplacian row start, first membrane entry}
  double* VR;                 // [Ntot][KN_REC]
  double* csol;               // [K-1][Ntot]  solver output c (block order handled by offsets)
  double* fsrc;               // [K-1][N_0] optional ECS source term (NULL when unused)
  const int* cells;           // [nctot][NV] global vertex ids
  // sliced ELL of (row, incident cell) pairs
  const int64_t* sl_ptr;      // [4*nblocks+1] entry offsets (multiples of KN_SLICE); slice 4*b + w
                              // holds the 64/lpr rows blk_row0[b] + (64/lpr)*w .. of block b; lane
                              // r*lpr + j of step p reads pair p*lpr + j of row r of the slice
  const uint32_t* pair_sl;    // simplices: NV x uint8 slots of the cell's vertices in the row's Laplacian
                              // segment (byte 0 = the row's own vertex); 0xFFFFFFFF marks padding.  Lattice tetrahedra
                              // (tet_tab != NULL): slots are 5 bits, bits 5-7 of byte 0 hold the cell's shape, bits
                              // 5-6 of bytes 1-3 the canonical vertex number of the vertex in that byte
  const double* tet_tab;      // lattice tetrahedra of a uniform grid (knpemi_create): [8 shapes][4][4] gradient dot
                              // products of the canonical vertices, then [8] cell volumes; NULL otherwise
  const int* pair_cell;       // hexahedra: cell*8 + local index, -1 = padding
  const uint32_t* pair_slots; // hexahedra: 8 x uint8 slots in two consecutive words per entry
  // EMI CSR (monolithic) and Laplacian-pattern CSR (KNP blocks share it per sub-domain)
  const int* rowptr; const int* colind; const uint8_t* lapoff;
  const int* rowptrL; const int* colindL;
  double* A_emi; double* P_emi; double* b_emi;
  double* A_knp; double* b_knp;          // block order (sub, ion)
  const int* krowptr; const int* kcolind; // monolithic block-diagonal KNP pattern (Krylov solve, export)
  int64_t nnz, nnzL;
  // membrane
  const int* mptr;            // [M+1]
  const int* mentry;          // facet*8 + local vertex a
  const uint64_t* mslots;     // bytes 0..3 own-side slot of col b, bytes 4..7 other-side slot
  const int* mrow;            // [M] global row of each membrane row
  // per membrane entry e = (row, facet, local vertex): model slot (-1: none), Q dofs of the facet, facet-mass row
  const int* me_model;        // [E]
  const int* me_q;            // [E][NF]
  const double* me_mass;      // [E][NF], filled once by membrane_mass_kernel
  const int* gam_pos;         // [nftot][2 sides][NF] -> entry e
  const int* fe; const int* fi; const int* fq;   // [nftot][NF] global ids
  const int* fmodel;          // [nftot] global model slot or -1
  const int* q2e; const int* q2i;                // [NQtot] global vertex ids
  const double* P_mass;       // [nnz - pmass0] static ICS mass entries of P_emi (rows of the cell sub-domains)
  int64_t pmass0;             // rowptr[first cell-side row]
  double* gam_e;              // [E][K-1] membrane partial integrals of b_knp, in entry order
  double* gpre;               // [E][(1 + NF)(K-1)] early form: phi-independent part + facet matrix (knp_membrane_pre_kernel)
  double* phiM;               // [NQtot]
  double* Ich;                // [n_model_slots][K][stride NQtot] (indexed by global q)
  int M;
  // membrane quadrature tables (degree 6): weights and shape values
  const double* qtab;         // [nq] weights, then [nq][NF] shape values, then (quads) [nq][NF][2] derivatives
  int nq_gamma;
};

struct KnOdeModel {
  int bound = 0, sub = 0, model_id = -1, n_states = 0, n_params = 0, nq = 0;
  double* d_states = nullptr;   // [n_states][nq]
  double* d_params = nullptr;   // [n_params][nq]
  uint8_t* d_mask = nullptr;    // [nq] or NULL
  int n_stim = 0;
  int stim_idx[8];
  double stim_val[8];
  unsigned long long* d_stats = nullptr; // [n_stat_blocks][3]: rhs evals, steps, failures per workgroup of the sweep
  int n_stat_blocks = 0;
  unsigned long long* d_stamps = nullptr;   // diagnostic phase stamps (KNPEMI_ODE_STAMPS)
  // model compiled at bind time from the plug-in's HIP source (kernels_rtc.hip); NULL for the shipped models
  void* rtc_module = nullptr;
  void* rtc_function = nullptr;
  int rtc_lanes = 1;
};

// blocks of the dense coarsest-level inverse, passed to the kernels by value: block b covers the unknowns start[b] ..
// start[b] + size[b] - 1, its row-major size[b] x size[b] inverse begins at off[b] (doubles)
struct KnDenseBlocks { int nb = 0; int start[8] = {0}; int size[8] = {0}; int off[8] = {0}; };

// geometry of the cells of a uniform hexahedral mesh (edge vectors e_t = x[1 << t] - x[0] of every cell): g = |det J| J^-1 J^-T,
// J = [e_0 e_1 e_2]; passed to the hexahedral row kernels by value
struct KnHexGeo { double g[3][3]; double det; int skew; };

// Algebraic multigrid hierarchy (kernels_amg.hip)
struct KnAmgCsr { int n = 0, m = 0, nnz = 0; int* rp = nullptr; int* ci = nullptr; double* v = nullptr; };
struct KnAmgLevel {
  int n = 0, nc = 0;                 // size, size of the next coarser level (0: dense coarsest level)
  int avg_row = 0, p_row = 0, r_row = 0;
  double omega = 0.0;                // Jacobi damping 4 / (3 rho)
  KnAmgCsr A, P, R;
  // merged transfer operators of the fused cycle (kernels_fused.hip): Rm = R (I - omega A D^-1), Pm = (I - omega D^-1 A) P,
  // built from the operator the hierarchy was set up with; frozen_v: that operator's values on the finest level (the
  // coarser levels' A.v are frozen copies already)
  KnAmgCsr Rm, Pm;
  double* frozen_v = nullptr;
  double* dinv = nullptr;
  // explicit inverse on the coarsest level as up to 8 dense diagonal blocks (amg_host.h: dense_inverse_blocks; the
  // aggregates of every level are numbered component by component, so the independent ion systems are contiguous ranges)
  double* dense_inv = nullptr;
  KnDenseBlocks dense_blk;
  double *x = nullptr, *r = nullptr, *t = nullptr;
};
struct KnAmgAsync;     // a rebuild running on a host thread (kernels_amg.hip)
struct KnAmg {
  KnAmgAsync* async = nullptr;
  bool rebuild_wanted = false;       // the hierarchy has aged (iteration count doubled): rebuild it in the background
  int solves = 0;                    // solves with this hierarchy's system (KNPEMI_AMG_REBUILD_EVERY test hook)
  std::vector<KnAmgLevel> lev;
  std::vector<void*> allocs;
  bool built = false, singular = false;
  bool negative_strength = false;    // strength of connection from -a_ij only (classical) instead of |a_ij|
  int n = 0;
  double theta = 0.08;               // strength threshold
  double op_complexity = 1.0;
  bool want_fused = false;           // also build the merged operators of the fused cycle (single rank, point smoother)
  bool fused_ok = false;             // ... and they exist: every level but the last has Rm / Pm, the last one a dense inverse
  int its_ref = -1;                  // iterations of the first solve after the build (rebuild trigger)
  int its_last = 4;                  // iterations of the last fused solve: size of the next solve's first chunk
  int builds = 0;
  // Optional aggregates of the finest level (auxiliary-space variant, DG systems: the broken dofs of a (sub-domain,
  // mesh vertex) form one aggregate, so the first coarse level is the continuous P1 space of the sub-domains and the
  // strength-based aggregation only starts there).  first_na == 0: aggregate every level by strength.
  std::vector<int> first_agg;
  int first_na = 0;
  // split_first: the given aggregates are split into the connected components of their strong couplings (-a_ij >=
  // split_theta sqrt(a_ii a_jj) between two dofs of ONE given aggregate): the interior penalty of a stretched cell
  // ties coincident dofs together across the facets along the long direction a hundred times more strongly than across
  // the others, and the low-energy error of the DG systems is continuous only across the former.
  bool split_first = false;
  bool positive_conflict = false;    // aggregation keeps strongly positively coupled unknowns apart (aggregate_apart)
  // Block-smoothed hierarchies (DG): the levels below the finest run through the merged transfer operators of the fused
  // cycle (kn_fused_subcycle): sub_fused asks for them at set-up, sub_fused_ok says they exist (dense coarsest level)
  bool sub_fused = false, sub_fused_ok = false;
  bool want_cycle = false, cycle_ok = false;   // point-Jacobi hierarchy of a rank's diagonal block: the whole cycle that way
  double* zero_sc = nullptr;         // 32 zeroed doubles: the "not done" flag the fused kernels look at
  double filter_theta = 0.0;         // > 0: prolongator smoothing with the filtered operator (weak entries lumped)
  bool first_tentative = false;      // the prolongator of the given aggregates is not smoothed
  double split_theta = 0.1;
  // Optional block-Jacobi smoother on the finest level: `block` consecutive unknowns (the dofs of a DG cell) form a
  // block whose inverse is refreshed from the current values before every solve (kn_amg_refresh).  0: point Jacobi.
  int block = 0;
  double* binv = nullptr;            // [n / block][block][block]
  double omega_block = 1.0;          // damping 4 / (3 rho(B^-1 A))
};
void kn_amg_free(KnAmg& G);
void kn_amg_async_join(KnAmg& G);    // ends (waits for) a background rebuild; before the hierarchy or its handle goes away
struct knpemi_handle;
int kn_amg_setup(knpemi_handle* h, KnAmg& G, int n, const int* d_rowptr, const int* d_colind, const double* d_vals,
                 bool singular, const uint8_t* h_owned = nullptr);
// G.rebuild_wanted: start the rebuild of an aged hierarchy on a host thread from a snapshot of the operator, or swap a
// finished one in; G stays usable throughout
int kn_amg_rebuild_step(knpemi_handle* h, KnAmg& G, int n, const int* d_rowptr, const int* d_colind, const double* d_vals,
                        bool singular);
int kn_amg_apply(knpemi_handle* h, KnAmg& G, const double* vals, const double* dinv0, const double* r, double* scratch,
                 double* out);
int kn_amg_refresh(knpemi_handle* h, KnAmg& G, const double* vals);   // block inverses of the current finest operator
// fused solver loops (kernels_fused.hip): the iteration of kn_solve_emi / kn_solve_knp in 6 / 13 launches
struct KnFusedSys {
  int n; const int* rowptr; const int* colind; const double* vals;   // the system of the current time step
  double* sc; double* work;   // the solver's device scalars and vector workspace (kernels_krylov.hip layout)
  size_t N;                   // stride of the workspace vectors
};
int kn_fused_subcycle(knpemi_handle* h, KnAmg& G, int l0, const double* r0 = nullptr, double* out0 = nullptr);
// pre / post: what the caller has to run before the first residual / after the last iteration (both become part of the
// captured graph of a chunk; post must be idempotent, it follows every chunk)
int kn_fused_cg(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit,
                int* iters, double* rr, double* bb, double* phi, int phi_stride);
int kn_fused_bicgstab(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit,
                      int* iters, double* rr, double* bb, const std::function<int()>& pre, const std::function<int()>& post);
int kn_fused_gmres(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit,
                   int* iters, double* rr, double* bb, const std::function<int()>& pre, const std::function<int()>& post);
void kn_fused_graphs_free(knpemi_handle* h);

// Distributed solves (knpemi_set_distributed)
struct KnDist {
  bool on = false;
  std::vector<uint8_t> h_owned_emi, h_owned_knp;   // host masks in the unknown order of the two systems
  uint8_t* d_owned_emi = nullptr;
  uint8_t* d_owned_knp = nullptr;
  double* d_red = nullptr;                         // caller's reduction buffer (>= 8 doubles)
  knpemi_allreduce_fn allreduce = nullptr;
  knpemi_halo_fn halo = nullptr;
  void* ctx = nullptr;
  double n_owned_global = 0.0;                     // owned EMI unknowns summed over the ranks
  // Coarse space of the distributed EMI preconditioner (knpemi_set_distributed_coarse): hat functions over k slices (along
  // the longest axis) of every sub-domain of every rank.  nc = world * n_sub * (k + 1) <= KN_COARSE_MAX.
  int rank = -1, world = 0, nc = 0, nl = 0, k = 1;   // nl = n_sub * (k + 1) local nodes, nc = world * nl
  bool coarse_built = false;
  double* d_coarse_inv = nullptr;                  // [nc][nc] (A_c + alpha 1 1^T)^-1, the same on every rank
  double* d_coarse_z = nullptr;                    // [nl] this rank's coarse corrections
  int* d_agg_of = nullptr;                         // [Ntot] lower local node of every unknown, -1 for ghosts
  double* d_agg_w = nullptr;                       // [Ntot] its weight in the upper node
  int* d_agg_ptr = nullptr;                        // [nl + 1]
  int* d_agg_idx = nullptr;                        // unknowns of every node ...
  double* d_agg_wt = nullptr;                      // ... and their weights
};
#define KN_COARSE_MAX 64
#define KN_COARSE_OFF 8                            // coarse vector sits behind the 8 scalars of the reduction buffer

// ghost refresh of a solver vector without Python (comm_rccl.hip)
struct KnVecPlan {
  bool set = false;
  const int32_t* send_idx = nullptr; const int32_t* recv_idx = nullptr;
  int n_send = 0, n_recv = 0;
  double* send_buf = nullptr; double* recv_buf = nullptr;
  std::vector<int32_t> peer;
  std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;
};

struct knpemi_handle {
  int device = 0;
  hipStream_t stream = nullptr;          // main stream
  hipStream_t aux = nullptr;             // auxiliary stream (EMI matrix assembly beside the ODE sweep)
  hipStream_t aux2 = nullptr;            // second auxiliary stream (ODE sweeps of further membrane models)
  hipEvent_t ev_join2 = nullptr;
  hipEvent_t ev_pre_fork = nullptr, ev_pre = nullptr;   // early membrane integrals on the auxiliary stream
  int pre_pending = 0;
  hipStream_t cur = nullptr;             // stream the row-kernel launchers enqueue on (stream or aux)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int gdim = 0, cell_kind = 0, NV = 0, NF = 0, n_sub = 0, K = 0;
  std::vector<int> n_vert, n_cell, n_q, n_facet, n_models;
  std::vector<int> voff, coff, qoff, foff, moff;
  KnConsts consts{};
  KnConsts* d_consts = nullptr;        // device copy read by the kernels
  const void* d_lsoda_coef = nullptr;  // LsodaCoef tables (kernels_ode.hip)
  KnDev dev{};
  int have_params = 0;
  int lpr = 1;                         // lanes per row of the row kernels (1, 2, 4 or 8)
  bool hex_affine = false;             // every hexahedron is a parallelepiped (constant Jacobian)
  bool tet_uniform = false;            // tetrahedra: every cell a lattice tetrahedron of a uniform grid (dev.tet_tab)
  bool hex_uniform = false;            // ... and all of them are the same one: the geometry below is a constant of the mesh
  KnHexGeo hex_geo{};
  int lds_doubles_emi = 0, lds_doubles_knp = 0; // per-block LDS segment sizes (doubles)
  int lds_uniq_max = 0;                         // most distinct vertices touched by one row block
  std::vector<void*> allocs;  // everything hipMalloc'ed
  std::vector<void*> rtc_modules;   // hipModule_t of run-time compiled membrane models
  std::vector<KnOdeModel> ode; // [moff[n_sub]]
  // host copies of patterns for export
  std::vector<int> h_rowptr, h_colind, h_rowptrL, h_colindL;
  void* comm = nullptr;                              // RCCL communicator (comm_rccl.hip), NULL until knpemi_comm_init
  int comm_rank = 0, comm_world = 1;
  KnVecPlan vec_plan[2];                             // [KNPEMI_B_EMI], [KNPEMI_B_KNP]
  double* d_stage = nullptr; size_t stage_len = 0;   // staging buffer for strided field I/O
  double* kry = nullptr; size_t kry_n = 0;           // Krylov workspace (kernels_krylov.hip)
  int kry_ones_masked = 0;                           // the workspace's `ones` vector currently holds the ownership mask
  void* kry_pinned = nullptr;                        // pinned host buffer the solvers' scalars are read through
  // state of the fused loops published by the device into mapped host memory (kernels_fused.hip: publish_state)
  double* pub_host = nullptr; double* pub_host_dev = nullptr; unsigned long long* pub_count = nullptr;
  unsigned long long pub_expected = 0;
  KnBlockCols bcols;                 // block structure of a DG problem's systems (solver handle of knpemi_dg), else empty
  int spmv_lpr[2] = {0, 0};          // lanes per row of the Krylov SpMV of the two systems (from the average row length)
  double* fused_part = nullptr; size_t fused_part_n = 0;   // block partials of the dot products fused into the solver kernels
  double* guess_old[2] = {nullptr, nullptr};         // previous solutions (EMI, KNP) for knpemi_extrapolate_guess
  int guess_have[2] = {0, 0};                        // previous solutions stored so far (0, 1, 2)
  KnAmg amg_emi, amg_knp;
  // captured iteration bodies of the Krylov loops (kernels_krylov.hip); key = configuration they were captured for
  struct KnGraph { hipGraphExec_t exec = nullptr; uint64_t key = 0; };
  KnGraph graph_emi, graph_knp;
  // captured chunks of the fused loops (kernels_fused.hip: run_chunk_graph), keyed by everything their kernel arguments hold
  std::unordered_map<uint64_t, hipGraphExec_t> fused_graphs;
  // graph replay or direct launches for the chunks of the fused loops, decided per system from timed solves (choose_mode)
  struct FusedMode { bool decided = false, graph = true; int solves = 0; double t[2] = {0.0, 0.0}; int n[2] = {0, 0}; };
  FusedMode fused_mode[2];
  int pc_emi = KNPEMI_PC_AMG, pc_knp = KNPEMI_PC_AMG;
  int fuse_update = 0;                 // KNPEMI_OPT_FUSE_UPDATE
  int knp_min_it = 0;                  // KNPEMI_OPT_KNP_MIN_IT (ksp_min_it of the concentration solve, pdeSolver.py:101)
  int emi_norm_pre = 0;                // KNPEMI_OPT_EMI_NORM: 1 = CG tests the preconditioned residual norm (KSPCG's default)
  int knp_method = 0;                  // KNPEMI_OPT_KNP_METHOD: 0 BiCGStab on the true residual, 1 GMRES(30) as PETSc runs it
  double* gm_V = nullptr; size_t gm_n = 0;   // Krylov basis of the GMRES solve
  double* gm_state = nullptr;                // its Hessenberg column, rotations, triangular factor, dot-product partial sums
  bool plain_knp = false;              // the KNP solve's unknown order is dev.csol's own ([K-1][Ntot]; DG variant)
  int fuse_membrane = 0;               // KNPEMI_OPT_FUSE_MEMBRANE
  // The membrane-facet integrals of b_knp in gam_e are those of the current fields (formed by the launch that wrote the
  // potential back, kn_launch_emi_writeback_membrane) with this splitting flag: knpemi_assemble_knp skips the facet kernel.
  // Cleared by everything that changes an input of the integrals (concentrations, phi, phi_M, I_ch, parameters).
  bool gam_valid = false;
  int gam_split = 1;
  int fold_membrane = 1;               // KNPEMI_OPT_FOLD_MEMBRANE: form them in the write-back launch of the potential
  int emi_flags = 0;                   // flags of the last knpemi_assemble_emi (the splitting scheme the step runs with)
  int prof_stride = 1;                 // KNPEMI_OPT_PROFILE_STRIDE
  unsigned prof_count[16] = {0};
  int lds_gam_max = 0;                 // most membrane entries of one row block
  bool blocks_clustered = false;       // row blocks are clusters of row chunks (default), not consecutive rows
  KnDist dist;
  // per-kernel event profiling (knpemi_profile)
  uint32_t prof_mask = 0;
  std::vector<hipEvent_t> prof_ev[KNPEMI_N_KERNELS];  // begin/end pairs
  size_t prof_used[KNPEMI_N_KERNELS] = {};
};

// RAII bracket around one kernel launch; no-op unless the kernel's bit is set in prof_mask.
struct KnProfScope {
  knpemi_handle* h; int k; bool on;
  KnProfScope(knpemi_handle* h_, int k_) : h(h_), k(k_), on((h_->prof_mask >> k_) & 1u) {
    // KNPEMI_OPT_PROFILE_STRIDE: bracket every n-th launch only (an event pair around a kernel on the critical path
    // costs the step several microseconds)
    if (on && h->prof_stride > 1 && (h->prof_count[k]++ % h->prof_stride) != 0) on = false;
    if (!on) return;
    auto& v = h->prof_ev[k];
    if (h->prof_used[k] + 2 > v.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
      v.push_back(a); v.push_back(b);
    }
    (void)hipEventRecord(v[h->prof_used[k]], h->cur);
  }
  ~KnProfScope() {
    if (!on) return;
    (void)hipEventRecord(h->prof_ev[k][h->prof_used[k] + 1], h->cur);
    h->prof_used[k] += 2;
  }
};

// error plumbing ------------------------------------------------------------------------------
void kn_set_error(const std::string& msg);
#define KN_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      kn_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                   \
      return KNPEMI_EHIP;                                                                \
    }                                                                                    \
  } while (0)

struct OdeDev;
struct OdeArgs;
int kn_launch_ode_raw(hipStream_t st, int model_id, const OdeDev& dv, const OdeArgs& a, const void* coef);   // kernels_ode.hip
int kn_lsoda_coef_upload(void** out);
void kn_comm_destroy(knpemi_handle* h);   // comm_rccl.hip
int kn_comm_create(int device, int rank, int world, const char* id_bytes, size_t len, void** out);
void kn_comm_free(void* comm);
int kn_comm_sendrecv(void* comm, int world, int device, hipStream_t stream, const double* send_buf_dev, double* recv_buf_dev,
                     int n_parts, const int32_t* peer, const int64_t* send_off, const int64_t* send_cnt,
                     const int64_t* recv_off, const int64_t* recv_cnt);
int kn_gamma_quadrature(int NF, std::vector<double>* out);   // degree-6 membrane-facet rule (knpemi_api.hip)

// kernel launchers (kernels_*.hip) ------------------------------------------------------------
int kn_launch_emi_rows(knpemi_handle* h, int flags);
int kn_launch_knp_rows(knpemi_handle* h, int flags);
int kn_launch_knp_membrane_pre(knpemi_handle* h, int flags);
int kn_launch_emi_membrane_rhs(knpemi_handle* h, int flags);
int kn_launch_knp_membrane(knpemi_handle* h, int flags);
int kn_launch_emi_writeback_membrane(knpemi_handle* h, const double* x, const double* part, int np, double inv_n, double* mean_out);
int kn_launch_membrane_mass(knpemi_handle* h, int n_entries, const int* d_entry_row, double* d_out);
int kn_launch_ode_step(knpemi_handle* h, int slot, double t0, double dt, double rtol, double atol,
                       int flags, const int32_t* ion_param, int v_index);
int kn_launch_update_pde(knpemi_handle* h);
int kn_rtc_bind(knpemi_handle* h, KnOdeModel& m, int n_states, int n_params, const char* rhs_source);
int kn_rtc_launch(knpemi_handle* h, const KnOdeModel& m, const void* dev_view, size_t dev_bytes, const void* args,
                  size_t args_bytes, const void* coef);
int kn_solve_emi(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres);
int kn_solve_knp(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres);
int kn_extrapolate_guess(knpemi_handle* h, int which);
int kn_launch_knp_order(knpemi_handle* h, double* x, int to_blocks);
int kn_launch_knp_writeback_update(knpemi_handle* h, const double* x);
int kn_launch_halo(knpemi_handle* h, int kind, int pack, const int32_t* idx, int n, double* buf);
int kn_launch_field_scatter(knpemi_handle* h, const double* src, double* dst, int n, int dst_stride);
int kn_launch_field_gather(knpemi_handle* h, const double* src, int src_stride, double* dst, int n);
int kn_launch_trace(knpemi_handle* h, const double* ue, const double* ui, int sub, double* qe, double* qi);
int kn_launch_vec_index(knpemi_handle* h, double* vec, const int32_t* idx, int n, double* buf, int gather);
