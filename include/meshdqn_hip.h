/*
 * meshdqn_hip.h - C ABI of the MI355X (gfx950) hot path of MeshDQN.
 *
 * Drop-in boundary for the data-parallel hot path of BaratiLab/MeshDQN
 * (Env2DAirfoil rollout = Taylor-Hood IPCS Navier-Stokes solve + graph
 * Q-network forward).  The reference has no native code and no FFI: all of its
 * arithmetic sits behind Python calls into DOLFIN / PyG.  Each entry point
 * below names the reference call site (file:line in BaratiLab/MeshDQN) whose
 * arithmetic it replaces; the Python classes in `meshdqn_amd/` that keep the
 * reference's `FlowSolver` / `Env2DAirfoil` / `NodeRemovalNet` surfaces bind
 * these symbols through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch types.
 *   - every call returns int: 0 = ok, <0 = error (text via mdq_last_error()).
 *   - all array arguments marked "device" are DEVICE pointers borrowed from
 *     the caller (torch tensors); the library never frees them and performs no
 *     hidden allocation; work is enqueued on the given hipStream_t (passed as
 *     void*; NULL = default stream) and the call returns without a device sync.
 *     Scratch memory is ALWAYS the caller's: every entry point that needs some
 *     takes a `workspace` (+ its size in bytes) whose minimum size a
 *     `*_workspace_bytes` / `*_workspace_doubles` query returns (0: none needed;
 *     < 0: the sizes are beyond the kernels); launches that may overlap - other
 *     streams - need workspaces of their own.  (ABI 5 kept the tables of the
 *     large-mesh kernel instances in process-static slabs grown with hipMalloc
 *     inside the calls: that broke this rule and is gone.)
 *   - "host" arguments are ordinary host pointers.
 *   - not re-entrant per batch descriptor; independent descriptors may be used
 *     from independent streams / processes.
 *
 * Batch layout: B environments, every per-environment array padded to common
 * capacities (NV vertices, NT triangles, NE edges, N2 = NV+NE scalar P2 dofs,
 * NNZ2 / NNZ1 non-zeros of the P2 / P1 CSR patterns, NSE2 / NSE1 entries of
 * their SELL-64 layouts); actual sizes in the
 * per-environment count arrays.  Velocity vectors are component-interleaved:
 * u[dof][c], dof in [0,N2), c in {x,y}.
 */
#ifndef MESHDQN_HIP_H
#define MESHDQN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDQ_ABI_VERSION 7

/* The library is built with -fvisibility=hidden: the entry points below are its ONLY exported symbols. */
#if defined(__GNUC__)
#define MDQ_API __attribute__((visibility("default")))
#else
#define MDQ_API
#endif

/* ---- error handling ----------------------------------------------------- */
MDQ_API int mdq_abi_version(void);
MDQ_API const char* mdq_last_error(void);

/* ---- IPCS batch descriptor ---------------------------------------------- */
typedef struct mdq_ipcs_desc {
  /* sizes */
  int32_t B;            /* environments in the batch                         */
  int32_t NV, NT, NE;   /* per-environment capacities                        */
  int32_t N2;           /* NV + NE (scalar P2 dof capacity)                  */
  int32_t NNZ2, NNZ1;   /* CSR capacities of the P2 / P1 patterns            */
  int32_t NAF;          /* capacity of the airfoil (tag 1) facet list        */
  int32_t NSE2, NSE1;   /* SELL-64 entry capacities of the P2 / P1 operators */
  /* physics: flow_params / solver_params of FlowSolver (flow_solver.py:49-52,95) */
  double mu, rho, dt;
  /* Krylov controls (the reference uses MUMPS LU, flow_solver.py:150-151;
     these bound the error of the iterative replacement) */
  double rtol;          /* relative residual tolerance (preconditioned norm) */
  int32_t maxit_u, maxit_p, maxit_m;
  int32_t mode;         /* operator application: -1 auto (fastest that fits), -2 auto among the BITWISE REPRODUCIBLE
                           variants (everything but 3: what ground truths / re-simulations of thousands of steps run
                           on, deploy_dqn.py:262-269, Env2DAirfoil.py:111-125), 0 assembled SELL (global gather vectors),
                           1 assembled SELL (LDS gather vectors), 2 matrix-free LDS element tiles (bitwise
                           reproducible), 3 matrix-free with LDS fp64 atomics (fastest; round-off reproducible),
                           4 assembled SELL with TWO workgroups per environment (any mesh size, rows / cells / slices
                           dealt out over the team, agent-scope team barriers; auto picks it over 0 when 4 B <= CUs),
                           5 matrix-free element tiles with the vectors in GLOBAL memory (any mesh size: what auto takes for
                           meshes beyond the LDS-resident vectors of 2 / 3 when mf_scat / mf_tptr are given; bitwise reproducible),
                           7 the element tiles of 5 with TWO workgroups per environment (chunks dealt out alternately, one
                           accumulation vector per workgroup; bitwise reproducible run to run, a row's sum associated
                           differently from 5; what auto takes instead of 5 while 2 B <= CUs).  6 is not a mode. */
  /* per-environment counts, device int32[B] */
  const int32_t* nv;
  const int32_t* nt;
  const int32_t* ne;
  const int32_t* naf;
  /* mesh, device */
  const double* coords;        /* [B][NV][2]  (smoothed) vertex coordinates   */
  const int32_t* cell_dofs;    /* [B][6][NT]  P2 dofs per cell: 3 vertices (ascending), 3 edge dofs NVenv+edge */
  const int8_t* cell_outflow;  /* [B][NT]     local edge (0..2) lying on the outflow boundary, or -1 */
  /* CSR patterns + deterministic gather maps, device */
  const int32_t* rowptr2;      /* [B][N2+1]   */
  const int32_t* colidx2;      /* [B][NNZ2]   */
  const int32_t* asm2_ptr;     /* [B][NNZ2+1] non-zero <- element slots       */
  const int32_t* asm2_src;     /* [B][36*NT]  slot = cell*36 + i*6 + j        */
  const int32_t* rowptr1;      /* [B][NV+1]   */
  const int32_t* colidx1;      /* [B][NNZ1]   */
  const int32_t* asm1_ptr;     /* [B][NNZ1+1] */
  const int32_t* asm1_src;     /* [B][9*NT]   slot = cell*9 + i*3 + j         */
  /* SELL-64 layout of the same patterns (slices of 64 consecutive rows, column-major
     inside a slice, width = longest row of the slice, padding column = own row) */
  const int32_t* sl2_off;      /* [B][N2/64+2]  slice offsets in entries      */
  const int32_t* sl2_col;      /* [B][NSE2]   */
  const int32_t* sl1_off;      /* [B][NV/64+2] */
  const int32_t* sl1_col;      /* [B][NSE1]   */
  /* matrix-free tile maps (mode 2): chunks of 1024 consecutive triangles          */
  const int32_t* mf_scat;      /* [B][6][NT]  packed: dof | tile position << 12 | (outflow edge + 1) << 28 (word 0); tile entries ordered by (row, triangle) */
  const int32_t* mf_tptr;      /* [B][NCH][N2+1] row r owns tile entries [tptr[r], tptr[r+1]) of chunk c; NCH = (NT+1023)/1024 */
  const int32_t* g2_ptr;       /* [B][N2+1]   P2 dof <- element slots         */
  const int32_t* g2_src;       /* [B][6*NT]   slot = cell*6 + i               */
  const int32_t* g1_ptr;       /* [B][NV+1]   */
  const int32_t* g1_src;       /* [B][3*NT]   slot = cell*3 + i               */
  /* Dirichlet data (flow_solver.py:123-132), device */
  const uint8_t* bcu_flag;     /* [B][N2]  1 = velocity dof constrained (both components) */
  const double* bcu_gx;        /* [B][N2]  x-velocity value (y value is 0)    */
  const uint8_t* bcp_flag;     /* [B][NV]  1 = pressure dof constrained to 0  */
  /* outflow-facet term of F1 (`- dot(mu*nabla_grad(U)*n, v)*ds`, flow_solver.py:109) as a tiny row list
     for the matrix-free mode 3: row r owns entries [bo_ptr[t], bo_ptr[t+1]) with 2x2 blocks
     B^{cd} = int_facet phi_r (d_c phi_col) n_d ds; values written by mdq_ipcs_assemble */
  int32_t NBO, NBE;            /* capacities: outflow rows, entries                 */
  const int32_t* nbo;          /* device [B]  number of outflow rows                */
  const int32_t* bo_rows;      /* device [B][NBO]   P2 dof of the row               */
  const int32_t* bo_ptr;       /* device [B][NBO+1]                                 */
  const int32_t* bo_col;       /* device [B][NBE]   P2 dof of the column            */
  const int32_t* bo_src;       /* device [B][NBE]   cell*36 + i*6 + j (local test / trial index) */
  double* bo_val;              /* device [B][NBE][4] (out) xx, xy, yx, yy           */
  /* direct (substructured) pressure solver, written by the host after assembly
     (meshdqn_amd/pressure_direct.py; replaces the reference's MUMPS factorisation of A2,
     flow_solver.py:150-159).  pd_enabled = 0 -> Jacobi-CG on the SELL operator instead. */
  int32_t pd_enabled;
  int32_t NPART, NPW, NPF, NPGI, NPS, NPGK;   /* capacities of the arrays below */
  int32_t pcg_degree;          /* mode 3, Krylov pressure solve (pd_enabled = 0 or no factors): degree m of the Chebyshev
                                  polynomial preconditioner of the CG on the Jacobi-scaled P1 Laplacian (m - 1 extra
                                  operator applications per iteration, no extra reductions); 0 / 1 = Jacobi only;
                                  < 0 = two-level additive preconditioner (8 x 7 geometric aggregates, coarse matrix
                                  inverted in LDS): fewer operator applications (167 -> ~100 iterations on ys930).
                                  Modes 0 / 4 / 5 / 7 (the meshes beyond mode 3; pressure vectors in LDS or, beyond
                                  ~4 000 vertices, in the workspace): 0 = auto - Jacobi-CG below 2048 vertices, the two-level
                                  preconditioner with an O(n) aggregation from there (refined ys930, 3 322 vertices: 325 ->
                                  158 iterations; refined twice, 12 924 vertices: 833 -> 470) -, < 0 = two-level wherever its
                                  scratch fits, > 0 = Jacobi-CG */
  const int32_t* pd_hdr;       /* [B][4]  nI, nG, nparts, 0                       */
  const int32_t* pd_node;      /* [B][NV] node id of permuted position (interiors by subdomain, then separator) */
  const int32_t* pd_meta;      /* [B][NPART][6] q0, m, W offset, F offset, g, gidx offset */
  const int32_t* pd_rowblk;    /* [B][NV] subdomain of permuted interior row       */
  const double* pd_W;          /* [B][NPW]  inv(K[I_s,I_s]) column-major blocks    */
  const double* pd_F;          /* [B][NPF]  W_s K[I_s,G_s] column-major blocks     */
  const int32_t* pd_gidx;      /* [B][NPGI] separator-local column ids of F blocks */
  const double* pd_Sinv;       /* [B][NPS]  inverse Schur complement, column-major nG x nG */
  const int32_t* pd_gk_ptr;    /* [B][NV+1] CSR of K[G,I] in permuted interior numbering */
  const int32_t* pd_gk_col;    /* [B][NPGK] */
  const double* pd_gk_val;     /* [B][NPGK] */
  /* airfoil facets for the probes (probes.py:23-50), device */
  const int32_t* af_facets;    /* [B][NAF][2] (cell, local edge)              */
  /* assembled operators, device, written by mdq_ipcs_assemble */
  double* geom;                /* [B][5][NT]  Jinv00,Jinv01,Jinv10,Jinv11,|det| */
  double* A1;                  /* [B][NSE2][4] SELL, row-scaled velocity blocks xx,xy,yx,yy */
  double* Ms;                  /* [B][NSE2]   SELL, symmetrically scaled P2 mass (with BCs) */
  double* K1s;                 /* [B][NSE1]   SELL, symmetrically scaled P1 stiffness (with BCs) */
  double* lift1;               /* [B][N2][2]  A1_full[:,bc] g                 */
  double* lift3;               /* [B][N2][2]  M_full[:,bc] g                  */
  double* idiag1;              /* [B][N2][2]  1/diag(A1)                      */
  double* sdiagM;              /* [B][N2]     sqrt(diag(M_bc))                */
  double* sdiagK;              /* [B][NV]     sqrt(diag(K1_bc))               */
  /* state, device */
  double* u_n;                 /* [B][N2][2]  */
  double* p_n;                 /* [B][NV]     */
  /* workspace, device: at least mdq_ipcs_workspace_doubles() doubles */
  double* work;
  int64_t work_doubles;
  /* mode 5 (optional; NULL: the row owners walk mf_tptr): per chunk the rows its 1024 triangles touch, ascending - entry =
     (row, first tile position | number of tile entries << 16) - so that a row phase visits ~2 500 touched rows instead of
     all N2 (12 924 on the refined ys930) */
  const int32_t* mf_rlist;     /* [B][NCH][NRL][2] */
  const int32_t* mf_rcnt;      /* [B][NCH] touched rows of the chunk */
  const int32_t* mf_lpos;      /* [B][6][NT] optional (NULL: off), with mf_rlist: position of dof i of triangle e in its chunk's row
                                  list (bits 0-15) | tile position (bits 16-31, = the plain mf_scat value): the chunk's input rows
                                  are then staged in LDS through the row list instead of gathered per triangle from global memory
                                  (needs NRL rows of 16 bytes behind the 96 KB tile: NRL <= ~3900) */
  int32_t NRL, rl_flags;             /* rl_flags = 1: bit 31 / bit 30 of a list entry's row word mark the FIRST / LAST chunk that
                                        touches the row, and every row < n2 is touched by some chunk (ABI 5, round 4) */
  /* ABI 7 (round 6), optional (NULL: off): [B] STICKY status words of the environments, OR-ed into by mdq_ipcs_evolve and never
     cleared by the library (the caller zeroes them).  MDQ_IPCS_TEAM_TIMEOUT: an IPCS step of the two-workgroup operator modes
     (4 / 7) was abandoned because a team barrier timed out - the partner workgroup was not resident (another process or stream
     held its CU).  That step and the later steps of the launch report NaN drag / lift; u_n / p_n keep the last completed step. */
  int32_t* status;
} mdq_ipcs_desc;
#define MDQ_IPCS_TEAM_TIMEOUT 1

/* doubles of workspace needed for a descriptor with the given capacities */
MDQ_API int64_t mdq_ipcs_workspace_doubles(int32_t B, int32_t NV, int32_t NT, int32_t NE);

/*
 * Assemble the three IPCS operators for every environment of the batch.
 * Replaces `SystemAssembler(a1,L1,bcu).assemble(A)` x3 and the UFL forms
 * (flow_solver.py:98-144; same for remesh in DEPLOY mode, :268-316):
 * per-triangle geometry + P2/P1 element matrices -> CSR values by a
 * deterministic gather, symmetric Dirichlet elimination, Jacobi scaling.
 */
MDQ_API int mdq_ipcs_assemble(const mdq_ipcs_desc* d, void* stream);

/*
 * A new mesh in the same descriptor (FlowSolver.remesh, flow_solver.py:233-359: the solvers restart without history): the
 * counters of the extrapolated initial guesses inside the workspace back to zero (the vectors behind them are not read
 * while their counter is zero), and - when `iters` is not NULL - the [B][3] iteration counters too.  Replaces a fill of
 * the whole workspace in front of mdq_ipcs_setup_matfree.
 */
MDQ_API int mdq_ipcs_reset_history(const mdq_ipcs_desc* d, int32_t* iters, void* stream);

/*
 * Operator setup of the MATRIX-FREE path only (mode 3, CG pressure solver) - what `FlowSolver.remesh` would have to
 * redo after every vertex removal (flow_solver.py:268-339 runs it in DEPLOY mode only): per-triangle geometry, the
 * outflow-row blocks, Jacobi diagonals and Dirichlet lifting vectors of A1 / M accumulated row-wise from the element
 * matrices (no global pattern), the scaled P1 Laplacian in SELL-64.  Needs from the descriptor: sizes, coords,
 * cell_dofs, cell_outflow, g1_*, g2_*, sl1_off/sl1_col, bcu_*, bcp_flag, bo_* ; writes geom, bo_val, lift1, lift3,
 * idiag1, sdiagM, sdiagK, K1s.  The pattern pointers (rowptr*, colidx*, asm*, sl2*) and A1/Ms may be NULL; such a
 * descriptor is valid for mdq_ipcs_evolve with mode = 3 and pd_enabled = 0 only.
 */
MDQ_API int mdq_ipcs_setup_matfree(const mdq_ipcs_desc* d, void* stream);

/*
 * ABI 7 (round 6).  The tile maps of the element-tile operator modes (5 / 7) for meshes that exist only on the device - the
 * coarsened meshes of the S3 env step, i.e. what `FlowSolver.remesh` (flow_solver.py:233-359) would derive after every
 * `Env2DAirfoil._remove_vertex` - built from the dof <- element-slot lists mdq_env_topology emits (g2_ptr, g2_src; needs nv, nt,
 * ne as well).  WRITES the arrays the descriptor's mf_rlist [B][NCH][NRL][2], mf_rcnt [B][NCH] and mf_lpos [B][6][NT] point at
 * (device, caller-owned; NCH = ceil(NT / 1024); NRL = row capacity of a chunk's list, <= 4064 for the LDS stage of the kernels):
 * bit for bit what `IpcsBatch` builds on the host for the same cells and dof numbering, with the first / last chunk bits of
 * rl_flags = 1.  mf_scat / mf_tptr stay NULL: mdq_ipcs_evolve takes the row lists + packed local maps alone.  An environment
 * whose maps cannot be built (a chunk that touches more than NRL rows, a dof without a cell) gets mf_rcnt[b][0] = -1 and keeps
 * the path through the element scratch and the dof <- slot lists; `status` (optional, device int32 [B]): 0 / 1 per environment.
 * Limits: NT <= 8192, N2 <= 16384, tables (NCH + 4) * N2 bytes <= 160 KB of LDS.
 */
MDQ_API int mdq_ipcs_build_tile_maps(const mdq_ipcs_desc* d, int32_t* status, void* stream);

/*
 * ABI 7 (round 6).  Spatial order for the cells of a PRIVATE copy of B meshes (the flow engine's input set of the S3 step, in
 * front of mdq_env_topology + mdq_ipcs_build_tile_maps): the cells of every mesh are sorted along a Morton curve of their
 * centroids (ties by cell id: the order is a function of the mesh alone) and `cells` [B][NT][3] and - when not NULL - `cell_dofs`
 * [B][6][NT] (another engine's dofs of the same cells) are permuted alike, in place.  Chunks of 1 024 consecutive triangles then
 * share their rows (~2 500 touched rows per chunk instead of ~5 000 in the order refinement and cavity re-triangulations leave):
 * what the tile maps need.  The reference has no counterpart: DOLFIN's assembly order is the mesh file's cell order
 * (flow_solver.py:123-144) and only the summation order of the element loops depends on it.  NT <= 8192.
 */
MDQ_API int mdq_flow_sort_cells(int32_t B, int32_t NV, int32_t NT, const double* coords, const int32_t* nv, const int32_t* nt,
                                int32_t* cells, int32_t* cell_dofs, void* stream);

/*
 * Advance every environment by `nsteps` IPCS time steps.
 * Replaces `FlowSolver.evolve()` (flow_solver.py:362-396): per step three
 * right-hand-side assemblies, three linear solves (BiCGStab / CG / CG in place
 * of the MUMPS back-substitutions), u_n/p_n update and the drag / lift probes
 * (probes.py:23-50).
 *   drag, lift : device double[B][nsteps]   (accumulated_drag / accumulated_lift)
 *   iters      : device int32[B][3] or NULL; Krylov iterations of the three
 *                solves are ADDED to it (velocity, pressure, correction)
 * State carried between calls: u_n, p_n and (mode 3) the tentative-velocity history in `work` from which the initial
 * guess of the velocity solve is extrapolated.  Zero `work` whenever u_n / p_n are replaced from outside or the mesh
 * changes (stale history only costs iterations, never accuracy: every solve still runs to `rtol`).
 */
MDQ_API int mdq_ipcs_evolve(const mdq_ipcs_desc* d, int32_t nsteps, double* drag, double* lift,
                    int32_t* iters, void* stream);

/*
 * Same as mdq_ipcs_evolve (`FlowSolver.evolve`, flow_solver.py:362-396) for the three-kernel mode 3, with HIP events recorded on `stream` around every
 * kernel launch; the accumulated durations (milliseconds over all nsteps) of the velocity / pressure /
 * correction kernels are returned in host array kernel_ms[3].  Synchronises the stream (measurement aid).
 */
MDQ_API int mdq_ipcs_evolve_timed(const mdq_ipcs_desc* d, int32_t nsteps, double* drag, double* lift,
                          int32_t* iters, void* stream, double* kernel_ms);

/*
 * Drag and lift of given fields on every environment's current mesh.
 * Replaces `DragProbe.sample(u,p)` / `LiftProbe.sample(u,p)` (probes.py:23-50)
 * as used by `Env2DAirfoil.calculate_reward` (Env2DAirfoil.py:390-394).
 *   u : device double[B][nfields][N2][2], p : device double[B][nfields][NV]
 *   drag, lift : device double[B][nfields]
 */
MDQ_API int mdq_probe_forces(const mdq_ipcs_desc* d, int32_t nfields, const double* u, const double* p,
                     double* drag, double* lift, void* stream);

/*
 * The factorisation half of `LUSolver("mumps")` (flow_solver.py:150-159; repeated by the reference for every
 * coarsened mesh, :318-328) on the device: builds the substructuring factors pd_* of every environment's scaled,
 * boundary-eliminated pressure matrix (K1s, as written by mdq_ipcs_assemble / mdq_ipcs_setup_matfree) - recursive
 * coordinate bisection into 8 parts, vertex separator, dense inverses of the interior blocks and of the Schur
 * complement by Gauss-Jordan in LDS.  The pd_* arrays of `d` must be allocated with capacities NPART >= 8,
 * NPW >= 8 * 112^2, NPF >= 8 * 112 * 48, NPGI >= 8 * 48, NPS >= 112^2 + 8 * 48^2 (the inverse Schur complement in front, scratch behind it), NPGK >= 112 * 16 (written here although the
 * descriptor declares them const: they are this call's outputs).  status (device int32 [B], may be NULL): 0 ok;
 * < 0: the mesh exceeds those limits - its header says nparts = 0 and mdq_ipcs_evolve (pd_enabled = 1) runs the
 * Krylov pressure solve for that environment.
 */
MDQ_API int mdq_ipcs_factorize_pressure(const mdq_ipcs_desc* d, int32_t* status, void* stream);

/* ---- graph Q-network forward (airfoilgcnn.py:85-145 NodeRemovalNet, :170-209 AirfoilGCNN) ---- */
typedef struct mdq_gcn_level {
  int32_t type;          /* 0 = SAGEConv (mean aggr, root weight), 1 = GCNConv (self loops, sym. norm) */
  int32_t fin;           /* input features of this level */
  const float* w_l;      /* device [fin][C]: SAGE lin_l.weight^T / GCN lin.weight^T */
  const float* b;        /* device [C]: SAGE lin_l.bias / GCN bias */
  const float* w_r;      /* device [fin][C]: SAGE lin_r.weight^T (unused for GCN) */
  const float* pool_w;   /* device [C]: TopKPooling.weight */
} mdq_gcn_level;

typedef struct mdq_gcn_net {
  int32_t nlevels, C, fin0, out_dim;  /* conv/pool levels, conv width, input features, head outputs */
  double ratio;                       /* TopKPooling ratio */
  int32_t softmax;                    /* 1: softmax over the head outputs (NodeRemovalNet) */
  int32_t _pad;
  mdq_gcn_level levels[6];
  const float* lin1_w;   /* device [2C][128]  lin1.weight^T */
  const float* lin1_b;
  const float* lin2_w;   /* device [128][64] */
  const float* lin2_b;
  const float* lin3_w;   /* device [64][out_dim] */
  const float* lin3_b;
} mdq_gcn_net;

/*
 * Batched forward of B graphs (one workgroup per graph, whole network out of LDS; dense head on MFMA).
 * Replaces `NodeRemovalNet.forward(Batch)` (airfoilgcnn.py:85-145) / `AirfoilGCNN.forward(Batch)` (airfoilgcnn.py:170-209)
 * and the PyG layers they call (SAGEConv / GCNConv / TopKPooling / global max + mean pool, airfoilgcnn.py:30-41).
 *   x        device float [sum_nodes][fin0]     node features, graphs concatenated
 *   node_ptr device int32 [B+1]                 first node of every graph
 *   esrc/edst device int32 [sum_edges]          edge end points, LOCAL node ids (0..n_g-1), j -> i
 *   edge_ptr device int32 [B+1]
 *   emb      device float [B][2C]  (out)        graph embeddings (x1+x2+...)
 *   out      device float [B][out_dim] (out)    head outputs (softmax probabilities if net.softmax)
 *   NMAX / EMAX: upper bounds of nodes / edges per graph (sizes the LDS carve-up)
 */
MDQ_API int mdq_gcn_forward(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                    const int32_t* node_ptr, const int32_t* esrc, const int32_t* edst,
                    const int32_t* edge_ptr, float* emb, float* out, void* stream);

/*
 * The same forward with two optional device outputs (either may be NULL):
 *   perm   int32 [B][nlevels][NMAX]  TopKPooling's `perm` of every level (airfoilgcnn.py:96: kept node r of level l =
 *                                    node perm[r] of the level's input; entries past ceil(ratio n) are -1) - the INDEX
 *                                    work of the Q-path, which the parity tests hold bit-exact against the oracle
 *   status int32 [B]                 0 ok, -1 / -2: graph b has more nodes / edges than NMAX / EMAX (its outputs are NaN;
 *                                    the LDS carve-up is sized from those bounds, a larger graph is never staged)
 */
MDQ_API int mdq_gcn_forward_ex(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                       const int32_t* node_ptr, const int32_t* esrc, const int32_t* edst,
                       const int32_t* edge_ptr, float* emb, float* out, int32_t* perm, int32_t* status, void* stream);

/* The same forward on edge lists PADDED to EMAX slots per graph (graph b: esrc_pad / edst_pad [b * EMAX .. + edge_cnt[b])) -
 * the layout mdq_env_topology writes (edge_src / edge_dst / nedges): no offsets, no compaction in front of the Q-forward
 * of a device-resident env step. */
MDQ_API int mdq_gcn_forward_padded(const mdq_gcn_net* net, int32_t B, int32_t NMAX, int32_t EMAX, const float* x,
                           const int32_t* node_ptr, const int32_t* esrc_pad, const int32_t* edst_pad,
                           const int32_t* edge_cnt, float* emb, float* out, int32_t* perm, int32_t* status, void* stream);

/* ---- the learning step of the graph Q-network (airfoil_dqn.py:240-310 DataWorker.compute_gradients) ---- */
/* Offsets (in floats) of every trained parameter inside the flat gradient: the parameters of the module in
 * `parameters()` order, each in torch layout [out][in] - the message of the gradient all-reduce. */
typedef struct mdq_gcn_grad_layout {
  int32_t w_l[6], b[6], w_r[6], pool_w[6]; /* per level: lin_l.weight | lin.weight, lin_l.bias | bias, lin_r.weight (SAGE), pool weight */
  int32_t lin1_w, lin1_b, lin2_w, lin2_b, lin3_w, lin3_b;
  int32_t total;                           /* floats of the flat gradient (ALL parameters of the module) */
  int32_t _pad;
} mdq_gcn_grad_layout;

typedef struct mdq_gcn_train_desc {
  int32_t B, NMAX, EMAX;   /* graphs of the minibatch; bounds of nodes / edges per graph (size the LDS carve-up) */
  int32_t mode;            /* which side of the double-DQN loss this network is on (the reference's `select` toggle):
                              0: loss_b = huber(out[b][action_b] - (reward_b + gamma nonfinal_b max_j q_other[b][j]))
                              1: loss_b = huber(q_other[b][action_b] - (reward_b + gamma nonfinal_b max_j out[b][j])) */
  double gamma;
  const float* x;          /* graphs as for mdq_gcn_forward: the states (mode 0) / the next states (mode 1) */
  const int32_t* node_ptr;
  const int32_t* esrc;
  const int32_t* edst;
  const int32_t* edge_ptr;
  const float* q_other;    /* device [B][out_dim] outputs of the OTHER network (no gradient): on the next states (mode 0) / the states (mode 1) */
  const int64_t* action;   /* device [B] */
  const float* reward;     /* device [B] */
  const float* nonfinal;   /* device [B] 1 where the transition has a next state, else 0 */
  float* workspace;        /* device [B][mdq_gcn_train_workspace()] floats of scratch */
  float* partial;          /* device [B][layout.total]: per-graph gradients; zero-filled ONCE by the caller (the slots of
                              parameters the forward never uses are never written and must read 0) */
  float* grad;             /* device [layout.total] (out): flat gradient of the mean loss */
  float* loss;             /* device [1] (out): mean Huber loss (delta 1) over the minibatch */
  float* out;              /* device [B][out_dim] (out, may be NULL): head outputs of this network */
  mdq_gcn_grad_layout layout;
} mdq_gcn_train_desc;

/* floats of workspace per graph for mdq_gcn_train_step (-1: bad arguments) */
MDQ_API int64_t mdq_gcn_train_workspace(const mdq_gcn_net* net, int32_t NMAX, int32_t EMAX);

/*
 * Forward + double-DQN Huber loss + full backward of `net` over a minibatch, without autograd: replaces
 * `loss.backward()` in DataWorker.compute_gradients (airfoil_dqn.py:286-310) and the PyG backward passes of SAGEConv /
 * GCNConv / TopKPooling / global max + mean pool behind it.  One workgroup per graph (forward out of LDS, backward over
 * the rows TopKPooling kept), then a reduction of the per-graph gradients in graph order: bitwise reproducible.
 */
MDQ_API int mdq_gcn_train_step(const mdq_gcn_net* net, const mdq_gcn_train_desc* d, void* stream);

/* Parameters of the module (torch layout) into the layout the kernels read, all segments in ONE launch:
 * segment s is a [rows][cols] matrix written transposed ([cols][rows]); cols == 1: a plain copy of `rows` floats. */
#define MDQ_GCN_PACK_MAX 32
typedef struct mdq_gcn_pack_table {
  int32_t n, _pad;
  const float* src[MDQ_GCN_PACK_MAX];
  float* dst[MDQ_GCN_PACK_MAX];
  int32_t rows[MDQ_GCN_PACK_MAX], cols[MDQ_GCN_PACK_MAX];
} mdq_gcn_pack_table;
MDQ_API int mdq_gcn_pack(const mdq_gcn_pack_table* table, void* stream);

/* ---- device-resident replay memory + optimiser update (airfoil_dqn.py:48-67 ReplayMemory, :184-200 apply_gradients) ---- */
/*
 * Transition records (float32; the layout trainer.pack_transitions writes and the ranks all-gather), N*F = nf, EM edge slots:
 *   [ x(s) nf | x(s') nf | src(s) EM | dst(s) EM | src(s') EM | dst(s') EM | edges(s) | edges(s') | action | reward | done ]
 * rec_len = 2 nf + 4 EM + 5; a terminal transition has zeros for s'.
 *
 * mdq_replay_step: `ReplayMemory.push` for the B environments of a batched step, in two halves, because the state arrays
 * of the vector environment are rewritten by the next step: the CURRENT batched state (x [B][nf], padded edge lists
 * [B][EM] + nedges [B], as VecEnv2DAirfoil hands them out) becomes s of the records (base_cur + b) % capacity and s' of
 * the records (base_prev + b) % capacity, whose action / reward / done [B] (of the step that led to this state) are
 * filled in too.  base_cur or base_prev < 0: that half is skipped.  All pointers device.
 */
MDQ_API int mdq_replay_step(float* ring, int32_t rec_len, int32_t capacity, int32_t B, int32_t nf, int32_t EM, const float* x,
                    const int32_t* edge_src, const int32_t* edge_dst, const int32_t* nedges, int32_t base_cur,
                    int32_t base_prev, const int32_t* action, const double* reward, const uint8_t* done, void* stream);

/*
 * `ReplayMemory.sample` + the batching of DataWorker._get_data (airfoil_dqn.py:63-64,240-262): records idx[0..n) of the
 * ring -> the graph arrays of mdq_gcn_forward / mdq_gcn_train_step for the states (`_s`) and the next states (`_n`; a
 * terminal transition contributes its own state as a placeholder, nonfinal = 0), edge lists packed back to back.
 */
typedef struct mdq_replay_sample_desc {
  int32_t n, rec_len, nf, EM;
  const float* R;             /* ring */
  const int32_t* idx;         /* [n] record numbers */
  float* x_s;                 /* [n][nf] */
  float* x_n;
  int32_t* esrc_s;            /* [n * EM] packed */
  int32_t* edst_s;
  int32_t* esrc_n;
  int32_t* edst_n;
  int32_t* edge_ptr_s;        /* [n + 1] */
  int32_t* edge_ptr_n;
  int64_t* action;            /* [n] */
  float* reward;              /* [n] */
  float* nonfinal;            /* [n] */
} mdq_replay_sample_desc;
MDQ_API int mdq_replay_sample(const mdq_replay_sample_desc* d, void* stream);

/*
 * torch.optim.Adam.step (amsgrad False; weight decay added to the gradient) for up to 32 parameter tensors in one
 * launch: segment s updates param[s][0..len[s]) from grad / exp_avg / exp_avg_sq [offset[s] ..] (flat buffers laid out
 * like the flat gradient).  bias_correction{1,2} = 1 - beta{1,2}^step, computed by the caller.
 */
typedef struct mdq_adam_desc {
  int32_t n, _pad;
  float* param[MDQ_GCN_PACK_MAX];
  int32_t offset[MDQ_GCN_PACK_MAX], len[MDQ_GCN_PACK_MAX];
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  double lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2;
} mdq_adam_desc;
MDQ_API int mdq_adam_step(const mdq_adam_desc* d, void* stream);

/* Up to 8 buffer copies in one launch (no reference counterpart; the env step hands its meshes and the warm-start fields
 * to the flow engine with it): buffer t = rows[t] rows of row_bytes[t] bytes, consecutive rows src_stride_bytes[t] /
 * dst_stride_bytes[t] apart (all multiples of 4, pointers 4-byte aligned; device pointers, host arrays of them). */
MDQ_API int mdq_copy_strided(int32_t n, void* const* dst, const void* const* src, const int64_t* rows, const int64_t* row_bytes,
                     const int64_t* src_stride_bytes, const int64_t* dst_stride_bytes, void* stream);

/* Probe kernel for meshdqn_amd/streams.py (no reference counterpart): `wgs` workgroups that each hold `lds_bytes` of LDS and
 * spin for `ticks_100mhz` ticks of the 100 MHz wall clock - with more workgroups than CUs it keeps the dispatcher of its
 * hardware queue busy, which is how two HIP streams are tested for really running beside each other. */
MDQ_API int mdq_spin(int32_t wgs, int32_t lds_bytes, int64_t ticks_100mhz, void* stream);

/* A HIP stream restricted to the compute units of `mask` (nwords 32-bit words; bit i = compute unit i in the driver's
 * numbering, which goes round-robin over the XCDs and then over the shader engines of an XCD), and its release.  No
 * reference counterpart: meshdqn_amd/streams.py gives the env step's main chain and its flow leg disjoint halves of the
 * chip (both are chains of kernels with one workgroup per environment). */
MDQ_API int mdq_stream_create_cu_mask(const uint32_t* mask, int32_t nwords, void** stream);
MDQ_API int mdq_stream_destroy(void* stream);

/* ---- snapshot interpolation onto coarsened meshes (Env2DAirfoil.py:556-593, :515-522) ---- */
typedef struct mdq_interp_desc {
  int32_t B, S;            /* target meshes (environments), snapshots                         */
  int32_t NP, NP1;         /* capacities: target P2 points (vertices then edge midpoints), P1 points (vertices) */
  int32_t src_nv, src_nt, src_n2;
  int32_t gnx, gny;        /* uniform location grid over the source mesh                       */
  int32_t _pad;
  double x0, y0, inv_hx, inv_hy;
  const int32_t* npts;     /* device [B]   P2 target points per environment                    */
  const int32_t* np1;      /* device [B]   P1 target points (= vertices) per environment       */
  const double* points;    /* device [B][NP][2]  target coordinates                            */
  /* source (original, smoothed) mesh and its snapshots, device */
  const double* src_coords;     /* [src_nv][2]                                                 */
  const int32_t* src_cell_dofs; /* [6][src_nt]                                                 */
  const double* src_geom;       /* [5][src_nt] as written by mdq_ipcs_assemble                 */
  const int32_t* bin_ptr;       /* [gnx*gny+1] candidate cells per grid bin (ascending ids, never empty) */
  const int32_t* bin_cells;
  const double* src_u;          /* [S][src_n2][2]                                              */
  const double* src_p;          /* [S][src_nv]                                                 */
  /* outputs, device */
  double* out_u;                /* [B][S][NP][2]                                               */
  double* out_p;                /* [B][S][NP1]                                                 */
  int32_t* out_cell;            /* [B][NP] located source cell (may be NULL)                   */
  /* optional, device: [src_nt][6] = {x, y of the cell's vertex 0, the four Jinv entries of src_geom} - what the point
   * location needs of a candidate cell in ONE record, 16-byte aligned (NULL: gathered from src_cell_dofs / src_coords /
   * src_geom) */
  const double* src_cellrec;
  /* optional, device [B]: added to npts[b] (a caller that holds vertex and edge counts separately passes np1 as npts and
   * the edge counts here instead of launching an add) */
  const int32_t* npts_extra;
  /* optional (ABI 5, round 4), `sparse` != 0: only what the device-resident env step reads of the S x (P2 + P1) fields -
   *   vertices: every snapshot (state features, pressures of the force integrals);
   *   edge midpoints: the LAST snapshot only (the warm start of the flow leg; sparse = 2: not even that - a step without
   *   a flow leg), and every snapshot for the three edges of the cells that carry an airfoil facet (the force integrals of
   *   the reward, probes.py:23-50).  The other entries of out_u keep whatever they held.  Needs the topology engine's
   *   airfoil-facet list and cell dofs of the TARGET meshes. */
  const int32_t* af_facets;     /* device [B][NAF][2] (cell, local facet)                      */
  const int32_t* naf;           /* device [B]                                                  */
  const int32_t* cell_dofs;     /* device [B][6][NT] dof ids of the target meshes              */
  int32_t NT, NAF;
  int32_t sparse, _pad2;
} mdq_interp_desc;

/*
 * Interpolate S stored (u, p) snapshots of the ORIGINAL mesh onto the P2 / P1 dof points of B
 * coarsened meshes: point location (containing cell, else nearest = extrapolation) + basis evaluation.
 * Replaces `v_func.interpolate(original_u)` / `p_func.interpolate(original_p)` (Env2DAirfoil.py:556-568).
 */
MDQ_API int mdq_interpolate_snapshots(const mdq_interp_desc* d, void* stream);

/* ---- host-side mesh smoothing (DOLFIN Mesh.smooth, flow_solver.py:65-67,236-237) ---- */
/*
 * Gauss-Seidel centroid smoothing of interior vertices in index order, step
 * limited to half the smallest altitude of the vertex star.  Host arrays.
 *   coords [nv][2] in/out; nbr_ptr[nv+1], nbr[]: vertex -> neighbour vertices;
 *   vc_ptr[nv+1], vc[]: vertex -> cell*3+local; cells[nt][3]; on_boundary[nv].
 */
MDQ_API int mdq_smooth_host(double* coords, int32_t nv, const int32_t* cells, int32_t nt,
                    const int64_t* nbr_ptr, const int64_t* nbr, const int64_t* vc_ptr,
                    const int64_t* vc, const uint8_t* on_boundary, int32_t iterations);

/*
 * Batched host mesh engine of the environment step (std::thread over environments):
 * remove one interior vertex per environment (remove_idx[b] < 0: none), restore the Delaunay
 * triangulation of the remaining points by star re-triangulation + Lawson flips (= the reference's
 * global `scipy.spatial.Delaunay` + all-boundary-simplex filter, Env2DAirfoil.py:480-496), then
 * DOLFIN-style smoothing (flow_solver.py:236-237).  Host arrays, in place:
 *   coords [B][NV][2], cells [B][NT][3] (any orientation in; ascending vertex ids per cell out),
 *   nv[B], nt[B] (updated), status[B] (0 ok, <0: star/ear-clipping/flip failure, mesh unusable).
 */
MDQ_API int mdq_remesh_host(int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv,
                    int32_t* nt, const int32_t* remove_idx, int32_t smooth_iters, int32_t nthreads,
                    int32_t* status);

/*
 * Node features of the state graphs of B environments (Env2DAirfoil.get_state, Env2DAirfoil.py:282-290), including the
 * reference's two indexing quirks (features indexed by n_closest = rank in the removable list; velocity block = raw
 * reshape of the (S,N,2) array).  coords [B][NV][2], u [B][S][NP][2], p [B][S][NV] (f64), n_closest [B][N], nsel [B]
 * (rows >= nsel are zero) -> x [B][N][2+3S] (f32).  All device pointers.
 */
MDQ_API int mdq_state_features(int32_t B, int32_t N, int32_t S, int32_t NV, int32_t NP, const double* coords, const double* u,
                       const double* p, const int32_t* n_closest, const int32_t* nsel, float* x, void* stream);

/*
 * Edge lists of the state graphs (Env2DAirfoil.get_state, Env2DAirfoil.py:258-280) from the padded per-environment
 * arrays of mdq_env_topology to the packed form mdq_gcn_forward reads: the first edge_ptr[b+1] - edge_ptr[b]
 * entries of src_pad / dst_pad [B][EMAX] go to esrc / edst [edge_ptr[b] ...).  All device pointers.
 */
MDQ_API int mdq_compact_edges(int32_t B, int32_t EMAX, const int32_t* src_pad, const int32_t* dst_pad, const int32_t* edge_ptr,
                      int32_t* esrc, int32_t* edst, void* stream);

/* ---- control logic of the batched env step on the device (Env2DAirfoil.step, Env2DAirfoil.py:318-377; calculate_reward
 *      :380-428): with these entry points a rollout is one uninterrupted stream of launches - no read-back between the
 *      Q-network forward and the next vertex removal.  All array arguments are device pointers over B environments. ---- */

/*
 * Action selection + decoding.  q [B][N+1] (or NULL: `action` holds the actions already): the greedy action is the first
 * maximum of a row (torch.argmax, airfoil_dqn.py:208-209); environment b takes rand_action[b] instead where explore[b]
 * (the host draws both arrays from its own random streams ahead of time).  Decoding as in Env2DAirfoil.step: action N
 * shifts the N-closest window (offset[b] += 1), an action without a vertex behind it (>= nsel[b]) gives code 2, otherwise
 * rem[b] = coord_map[b][action] (else -1).  Outputs: action, rem, code [B].
 */
MDQ_API int mdq_env_act(int32_t B, int32_t N, const float* q, const uint8_t* explore, const int32_t* rand_action,
                const int32_t* nsel, const int32_t* coord_map, int32_t* offset, int32_t* action, int32_t* rem,
                int32_t* code, void* stream);

/* its[b] = iterations where a vertex was removed (rem >= 0) and the re-triangulation succeeded (rstat == 0), else 0:
 * the smoothing request of flow_solver.py:236-237 for mdq_smooth. */
MDQ_API int mdq_env_smooth_iters(int32_t B, const int32_t* rem, const int32_t* rstat, int32_t iterations, int32_t* its,
                         void* stream);

/*
 * Reward and terminal flag (Env2DAirfoil.calculate_reward, Env2DAirfoil.py:380-428, + the bookkeeping of step()):
 * new_drags [B][S] (interpolated snapshots' drags on the current mesh), gt_drag [S]; drag reward
 * 2 exp(-(2 ln 2 / threshold) |rel. error|_2) - 1 + time_reward (nv0 - nv); terminal when any relative drag error exceeds
 * the threshold, nv < goal_vertices nv0, the step count reaches `timesteps`, or the step failed (code 2: rstat != 0,
 * nsel < N, topology status != 0 - the latter also sets bit 0 of *err_flag); failed steps get negative_reward.
 * code / steps [B] in/out (steps restart at 0 for terminated environments when auto_reset); reward f64 [B], done u8 [B].
 */
MDQ_API int mdq_env_result(int32_t B, int32_t N, int32_t S, const double* new_drags, const double* gt_drag, const int32_t* nv,
                   int32_t nv0, const int32_t* rstat, const int32_t* topo_status, const int32_t* nsel, int32_t* code,
                   int32_t* steps, double threshold, double time_reward, double goal_vertices, int32_t timesteps,
                   double negative_reward, int32_t auto_reset, double* reward, uint8_t* done, int32_t* err_flag,
                   int32_t* nv_out /* optional [B]: copy of nv (the vertex counts of this step) */, void* stream);

/* mdq_restore_rows for the environments with mask[b] != 0 (device array): the in-place reset without a host-side list. */
MDQ_API int mdq_restore_rows_masked(int32_t n, void* const* dst, const void* const* src, const int64_t* row_bytes, int32_t B,
                            const uint8_t* mask, void* stream);

/*
 * The END of a device-resident env step in ONE launch (round 4; before: mdq_copy_strided + mdq_env_result +
 * mdq_restore_rows_masked + mdq_state_features, four launches of 5-9 us each on the latency chain of the step), one
 * environment per workgroup row:
 *   1. reward / terminal flag / codes / step counters exactly as mdq_env_result (Env2DAirfoil.py:380-428, :342-377); the step
 *      counters and codes are read from *_in and written to *_out (several workgroups of an environment read them);
 *   2. every row array t < n_rows (dst[t] laid out [B][row_bytes[t]]): the part [handover_off, + handover_bytes) of
 *      environment b's row is first copied to handover_dst[t] ([B][handover_bytes[t]]; NULL: none) - the meshes and warm-start
 *      fields the flow stream's IPCS leg of THIS step works on, taken before a reset rewrites them - and then, where the
 *      environment terminated and auto_reset is set, the row is overwritten with the cached initial row src[t] (NULL: the
 *      array is handed over only) - Env2DAirfoil.reset, Env2DAirfoil.py:102-129, in place;
 *   3. the node features of the NEXT state (mdq_state_features, Env2DAirfoil.get_state :244-290): computed from the
 *      environment's current arrays, or - for an environment that was just reset - the cached features x_init [N][2 + 3 S]
 *      of the initial state.
 * All pointers are device pointers; sizes / offsets are multiples of 4 bytes (16-byte aligned rows are copied 16 bytes
 * per lane).
 */
#define MDQ_FINISH_MAX_ROWS 16
typedef struct {
  int32_t B, N, S, NV, NP, n_rows;
  int32_t nv0, timesteps, auto_reset, _pad;
  double threshold, time_reward, goal_vertices, negative_reward;
  /* 1. result */
  const double* new_drags;     /* [B][S] */
  const double* gt_drag;       /* [S] */
  const int32_t* nv;           /* [B] */
  const int32_t* rstat;        /* [B] status of mdq_remesh */
  const int32_t* topo_status;  /* [B] status of mdq_env_topology (or NULL) */
  const int32_t* nsel;         /* [B] */
  const int32_t* code_in;      /* [B] codes of the action decoding (mdq_env_act / mdq_remesh_act) */
  int32_t* code_out;           /* [B] */
  const int32_t* steps_in;     /* [B] */
  int32_t* steps_out;          /* [B] (a different array) */
  double* reward;              /* [B] */
  uint8_t* done;               /* [B] */
  int32_t* err_flag;           /* [1] */
  int32_t* nv_out;             /* [B] or NULL */
  /* 2. hand-over + in-place reset */
  void* dst[MDQ_FINISH_MAX_ROWS];
  const void* src[MDQ_FINISH_MAX_ROWS];
  int64_t row_bytes[MDQ_FINISH_MAX_ROWS];
  void* handover_dst[MDQ_FINISH_MAX_ROWS];
  int64_t handover_off[MDQ_FINISH_MAX_ROWS];
  int64_t handover_bytes[MDQ_FINISH_MAX_ROWS];
  /* 3. features of the next state */
  const double* coords;        /* [B][NV][2] (one of the row arrays above) */
  const double* u;             /* [B][S][NP][2] */
  const double* p;             /* [B][S][NV] */
  const int32_t* n_closest;    /* [B][N] */
  const float* x_init;         /* [N][2 + 3 S] or NULL (then auto_reset environments get their features computed too - from
                                  the rows this launch has just restored: only valid with ONE workgroup per environment) */
  float* x;                    /* [B][N][2 + 3 S] */
  int32_t* arrive;             /* [B] arrival counters of the workgroups of an environment, ZERO before the first launch and
                                  left at zero by every launch: the rows among dst[] that ARE nv / nsel - inputs of the
                                  terminal decision every workgroup of the environment takes - are restored by the last
                                  workgroup to have read them.  NULL: one workgroup per environment */
} mdq_env_finish_desc;
MDQ_API int mdq_env_finish(const mdq_env_finish_desc* d, void* stream);

/*
 * mdq_env_act + mdq_remesh in one launch (the action decoding is the head of the removal kernel; same outputs as the two
 * entry points called one after the other): q / explore / rand_action / nsel / coord_map / offset / action / rem / code as
 * for mdq_env_act, the mesh arguments and status as for mdq_remesh.
 */
MDQ_API int mdq_remesh_act(int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv, int32_t* nt, int32_t N,
                   const float* q, const uint8_t* explore, const int32_t* rand_action, const int32_t* nsel,
                   const int32_t* coord_map, int32_t* offset, int32_t* action, int32_t* rem, int32_t* code, int32_t* status,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* edge_ptr [B+1] = exclusive prefix sums of nedges [B] (offsets of mdq_compact_edges / mdq_gcn_forward). */
MDQ_API int mdq_edge_ptr(int32_t B, const int32_t* nedges, int32_t* edge_ptr, void* stream);

/*
 * Env2DAirfoil.reset (Env2DAirfoil.py:102-129: mesh, snapshots and selection back to the initial ones) for a SUBSET
 * of the batched environments, in one launch: for each of the n (<= 16) device tensors dst[t], laid out [B][row_bytes[t]],
 * row idx[i] (i < n_idx) is overwritten with the cached initial row src[t].  dst / src / row_bytes are host arrays
 * of device pointers / sizes (4-byte aligned multiples of 4 bytes); idx is a device pointer.
 */
MDQ_API int mdq_restore_rows(int32_t n, void* const* dst, const void* const* src, const int64_t* row_bytes, int32_t n_idx,
                     const int32_t* idx, void* stream);

/*
 * DOLFIN `Mesh.smooth(n)` (flow_solver.py:65-67 and 236-237 after every remesh) for B meshes on the GPU: Gauss-Seidel
 * over the interior vertices in index order, each moved towards the centroid of its neighbours by at most half the
 * minimum altitude of its cells; boundary vertices (an incident edge with one owner) are fixed.  One workgroup per
 * mesh: a list schedule of the sweep's dependency graph, walked by one wave out of LDS (exact sequential semantics).  coords [B][NV][2] (in/out),
 * cells [B][NT][3], nv / nt [B], iterations [B] (0 = leave that mesh alone); all device pointers.
 * Capacity: NV <= 1024, NT <= 2048 out of LDS (workspace NULL / 0); up to NV <= 4096, NT <= 8192 through the level-scheduled
 * large-mesh kernel, whose tables live in `workspace` (device, 16-byte aligned, >= mdq_smooth_workspace_bytes(B, NV, NT));
 * up to NV <= 16384, NT <= 32768 (round 6) with the positions in `workspace` as well (-1 beyond that).
 */
MDQ_API int64_t mdq_smooth_workspace_bytes(int32_t B, int32_t NV, int32_t NT);
MDQ_API int mdq_smooth(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
               const int32_t* nt, const int32_t* iterations, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The same smoothing, faster: a Gauss-Seidel sweep in which every vertex takes the full step is the linear system
 * (D - L) x_new = U x_old: the sweeps run as blocked triangular solves (32-row blocks, inverses built once per launch),
 * every sweep is validated in parallel (was each update clearly a full step?), and a sweep that was not is redone from
 * its snapshot with the offending vertices on DOLFIN's exact (limited-step) update and a rank-1 correction of their block:
 * exact sequential semantics, results equal to mdq_smooth to round-off (different association).  A mesh beyond the
 * kernel's limits (more than 16 cells at a vertex) is handed to mdq_smooth's kernel.  Same arguments as mdq_smooth + a
 * device workspace of at least mdq_smooth_fast_workspace_bytes(B, NV) bytes, 16-byte aligned (block inverses,
 * per-environment diagnostics).
 */
MDQ_API int64_t mdq_smooth_fast_workspace_bytes(int32_t B, int32_t NV);
MDQ_API int mdq_smooth_fast(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
                    const int32_t* nt, const int32_t* iterations, void* workspace, int64_t workspace_bytes, void* stream);
/* Inside an env step (Env2DAirfoil._check_mesh -> flow_solver.remesh -> smooth(50), flow_solver.py:236-237): `iterations`
 * sweeps for the environments whose vertex removal succeeded (rem[b] >= 0 and rstat[b] == 0, the outputs of mdq_env_act /
 * mdq_remesh), none for the others - mdq_env_smooth_iters + mdq_smooth_fast in one launch. */
MDQ_API int mdq_smooth_fast_env(int32_t B, int32_t NV, int32_t NT, double* coords, const int32_t* cells, const int32_t* nv,
                        const int32_t* nt, const int32_t* rem, const int32_t* rstat, int32_t iterations, void* workspace,
                        int64_t workspace_bytes, void* stream);

/*
 * Diagnostics of mdq_smooth since the last reset (no reference counterpart): out64[s], s < 63 = speculative sweeps s
 * that met an update that was not clearly a full step and were redone by careful sweeps (see mdq_smooth.hip).
 * out64 is a HOST array of 64 int64 (may be NULL); reset != 0 zeroes the counters.  Synchronises the device.
 */
MDQ_API int mdq_smooth_stats(int64_t* out64, int32_t reset);

/*
 * Env2DAirfoil._remove_vertex (Env2DAirfoil.py:452-512) for B meshes on the GPU, WITHOUT the smoothing (mdq_smooth):
 * ear clipping of the removed vertex's star polygon + Lawson flips to the (unique) Delaunay triangulation = the
 * reference's global scipy Delaunay + all-boundary filter as a set of cells; vertex ids above the removed one shift
 * down; cells are written with ascending vertex ids.  remove_idx[b] < 0 leaves mesh b untouched.  status[b]: 0 ok,
 * -1..-4 star / boundary vertex / ear clipping failures, -11 non-manifold, -12 flip work list exhausted (mesh b is
 * untouched on failure).  All pointers are device pointers.  Capacity: NV <= 1024, NT <= 2048 with every table in LDS
 * (workspace NULL / 0); up to NV <= 4096, NT <= 8192 with the tables in `workspace` (device, 16-byte aligned, at least
 * mdq_remesh_workspace_bytes(B, NV, NT) bytes: 0 for the LDS instance, -1 beyond the kernels) and the edge hash in LDS; up to
 * NV <= 16384, NT <= 32768 (round 6) with the edge hash in `workspace` as well.
 */
MDQ_API int64_t mdq_remesh_workspace_bytes(int32_t B, int32_t NV, int32_t NT);
MDQ_API int mdq_remesh(int32_t B, int32_t NV, int32_t NT, double* coords, int32_t* cells, int32_t* nv, int32_t* nt,
               const int32_t* remove_idx, int32_t* status, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- optional outputs of mdq_env_topology_host: the index data of the matrix-free IPCS path (mode 3 with the CG
 *      pressure solver) on every coarsened mesh, i.e. what FlowSolver.__init__/remesh derive from the mesh
 *      (flow_solver.py:85-132,194-226) minus the assembled patterns.  Same layouts as the mdq_ipcs_desc fields of the
 *      same names; feed them (uploaded) to mdq_ipcs_setup_matfree + mdq_ipcs_evolve. ---- */
typedef struct mdq_ipcs_topo_out {
  int32_t NBO, NBE, NSE1;  /* capacities: outflow rows, outflow entries, SELL-64 entries of the P1 Laplacian */
  int32_t flow_only;      /* mdq_env_topology: != 0 skips `removable`, the polygon distances / N-closest window and the state
                             graph (nremovable / nsel / n_closest / coord_map / nedges / edge_* are left untouched): the run of
                             an engine that only feeds the IPCS step (the flow stream of the S3 env step) */
  int32_t* mf_scat;       /* [B][6][NT]  dof | (outflow_edge+1) << 28 (word 0) */
  int8_t* cell_outflow;   /* [B][NT]     local outflow facet or -1 */
  uint8_t* bcu_flag;      /* [B][NP] */
  double* bcu_gx;         /* [B][NP] */
  uint8_t* bcp_flag;      /* [B][NV] */
  int32_t* nbo;           /* [B] */
  int32_t* bo_rows;       /* [B][NBO] */
  int32_t* bo_ptr;        /* [B][NBO+1] */
  int32_t* bo_col;        /* [B][NBE] */
  int32_t* bo_src;        /* [B][NBE] */
  int32_t* g1_ptr;        /* [B][NV+1] */
  int32_t* g1_src;        /* [B][3*NT] */
  int32_t* g2_ptr;        /* [B][NP+1] */
  int32_t* g2_src;        /* [B][6*NT] */
  int32_t* sl1_off;       /* [B][NV/64+2] */
  int32_t* sl1_col;       /* [B][NSE1] */
  /* optional inputs of a `flow_only` run of mdq_env_topology (both NULL: the edges are numbered here): the cell dofs
   * [B][6][NT] and edge counts [B] that another engine's run derived from the SAME meshes - the edge numbering is taken
   * from them instead of being found again through the hash table */
  const int32_t* cell_dofs_in;
  const int32_t* ne_in;
} mdq_ipcs_topo_out;

/* ---- batched topology + N-closest selection + state graph (host arrays) ---- */
/* Optional second set of outputs of mdq_env_topology (device engine only; ABI 5, round 4): copies of the mesh and of its
 * edge numbering for ANOTHER engine that works on the same meshes on another stream - the flow leg of the S3 env step
 * (VecEnv2DAirfoil, flow_overlap): written by the topology kernel itself, so that the other stream can start as soon as this
 * launch has finished instead of waiting for a copy launch behind it.  A HOST struct holding device pointers. */
typedef struct mdq_topo_handover {
  double* coords;           /* [B][NV][2] rows < nv */
  int32_t* cells;           /* [B][NT][3] rows < nt */
  int32_t* nv;              /* [B] */
  int32_t* nt;              /* [B] */
  int32_t* cell_dofs;       /* [B][6][NT] */
  int32_t* ne;              /* [B] */
} mdq_topo_handover;

typedef struct mdq_env_topo_desc {
  int32_t B, NV, NT, NP, NAF, N, EMAX, npoly;   /* capacities; N = N_closest; npoly = airfoil polygon points */
  /* inputs */
  const double* coords;     /* [B][NV][2] */
  const int32_t* cells;     /* [B][NT][3]  ascending vertex ids per cell */
  const int32_t* nv;        /* [B] */
  const int32_t* nt;        /* [B] */
  const int32_t* offset;    /* [B]  do_nothing_offset (Env2DAirfoil.py:308) */
  const double* polygon;    /* [npoly][2]  airfoil boundary points of the ORIGINAL mesh in vertex order (:223-233) */
  /* outputs */
  int32_t* ne;              /* [B] */
  int32_t* cell_dofs;       /* [B][6][NT]  P2 dofs (edges numbered by first appearance) */
  double* points;           /* [B][NP][2]  P2 dof coordinates: vertices, then edge midpoints */
  int32_t* naf;             /* [B] */
  int32_t* af_facets;       /* [B][NAF][2] (cell, local edge) of the airfoil facets */
  int32_t* nremovable;      /* [B] */
  int32_t* nsel;            /* [B]  min(N, removable - offset) */
  int32_t* n_closest;       /* [B][N]  ranks inside the removable list (the reference indexes features with these) */
  int32_t* coord_map;       /* [B][N]  vertex id of every action */
  int32_t* nedges;          /* [B] */
  int32_t* edge_src;        /* [B][EMAX] */
  int32_t* edge_dst;        /* [B][EMAX] */
  double* edge_len;         /* [B][EMAX] edge_attr */
  const mdq_ipcs_topo_out* ipcs;  /* optional (NULL: skip) */
  const mdq_topo_handover* handover;  /* optional (NULL: none; ignored by the host engine) */
  void* workspace;          /* device, 16-byte aligned, >= mdq_env_topology_workspace_bytes(d) bytes: the tables of the
                               large-mesh kernel instance (NULL / 0 for meshes of up to 1024 vertices and for the host engine) */
  int64_t workspace_bytes;
} mdq_env_topo_desc;

/*
 * Per environment: unique edges / P2 dof map / dof coordinates, boundary + airfoil facets, `removable`
 * (flow_solver.py:75-78 quirk), polygon distances + argsort + N-closest window (Env2DAirfoil.py:220-241,
 * 293-315) and the state graph edges (:258-280).  status[b] = 0 ok, <0 capacity exceeded
 * (-1 NP, -2 NAF, -3 EMAX, -4 more than 2 outflow rows per row-owner thread, -5 NBO/NBE, -6 NSE1).
 */
MDQ_API int mdq_env_topology_host(const mdq_env_topo_desc* d, int32_t nthreads, int32_t* status);

/*
 * The same engine as a HIP kernel (one workgroup per environment, everything in LDS): every pointer of the descriptor
 * (and of d->ipcs, a HOST struct holding device pointers) and `status` are DEVICE pointers.  All output arrays are
 * bit-identical to mdq_env_topology_host's.  Capacity: NV <= 1024, NT <= 2048, NP <= 4096 with every table in LDS; up to
 * NV <= 4096, NT <= 8192, NP <= 16384 with the tables in d->workspace (mdq_env_topology_workspace_bytes: 0 for the LDS
 * instance, -1 beyond the kernels), npoly <= 256; up to NV <= 16384, NT <= 32768, NP <= 65536 (round 6: every table incl.
 * the edge hash and the coordinates in d->workspace), npoly <= 512.
 */
MDQ_API int64_t mdq_env_topology_workspace_bytes(const mdq_env_topo_desc* d);
MDQ_API int mdq_env_topology(const mdq_env_topo_desc* d, void* stream, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* MESHDQN_HIP_H */
