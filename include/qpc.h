/*
 * qpc.h -- C ABI of libqpcontrol_hip.so: batched whole-body-QP control on MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of tkoolen/QPControl.jl: one
 * `MomentumBasedController` tick (reference src/lowlevel/momentum.jl:41-81) for a BATCH of
 * independent robot states.  A Julia host `ccall`s these entry points (binding shown in
 * INTEGRATION.md and qpcontrol.jl_amd/julia/QPControlHIP.jl); the Python host mirror in
 * qpcontrol.jl_amd/ binds the same symbols through ctypes.
 *
 * Conventions
 *  - Plain C types only.  All functions return an int32 error code (QPC_OK == 0) unless noted;
 *    qpc_last_error() returns a thread-local description of the last failure.
 *  - Batched arrays are "column-major (dim x B)" in Julia terms: the record of QP b is
 *    contiguous, element (i, b) lives at [b*dim + i].  Julia `Matrix{Float64}(dim, B)` and numpy
 *    C-order arrays of shape (B, dim) both have exactly this layout.
 *  - Bodies are numbered 0..nbodies-1 in topological order (parent[i] < i); the world/root body is
 *    QPC_WORLD (-1).  Joint i is the joint between parent[i] and body i ("joint" arguments are
 *    therefore body indices).  Body frame i == the frame after joint i (RigidBodyDynamics'
 *    default frame of a URDF-parsed body).
 *  - Spatial vectors are [angular(3); linear(3)] (RigidBodyDynamics layout).
 *  - Handles are opaque, freed by *_destroy, usable from one host thread at a time.
 *  - Limits of the one-wavefront-per-QP kernels (qpc_finalize returns QPC_ERR_UNSUPPORTED beyond them):
 *    bodies <= 64, velocity coordinates nv <= 64, generic (non-pinned) tasks <= 63, tree depth <= 32
 *    (<= 16 without the hiprtc-specialised kernel), equality rows of the reduced QP (task rows that
 *    are not pins + 6 wrench-balance rows) <= 64, friction basis N <= 8, per-QP LDS slab <= 160 KiB.
 *    Exercised AT these limits (tests/test_valkyrie.py, MI355X): a Valkyrie-sized humanoid of 59 bodies, nv = 64, tree depth 15, eight
 *    contact points -- the reference's own integration tests (test/controller.jl:169-285) at their tolerances, 34 KB of LDS per QP,
 *    92 us per tick at B = 1 and 8.4 M ticks/s at B = 2 048 (profiles/r05_valkyrie_latency.json); beyond Atlas in other directions:
 *    44 bodies with 72 friction rays and 45 equality rows, and a tree of exactly 64 bodies and 64 velocity coordinates -- 65 bodies are
 *    refused -- (tests/test_large_mechanism.py).  The number of contact points has no limit of its own (the slab has): 33 points x 8 rays
 *    = 264 rho's against the oracle in tests/test_edge_contacts.py, 80 points (320 rho's, 83 KB of LDS) offline; the warm start's two
 *    64-bit masks hold at most 64 rho's (qpc_enable_warm_start is refused beyond).  The solver-level plug (qpc_qp_setup /
 *    qpc_qp_solve) takes n, m <= 256; the Valkyrie standing controller's reference-layout problem (171 x 206) runs in its sparse
 *    form at 0.13 ms (B = 1, median) with 148 KB of LDS -- what bounds that form is LDS: plan tables + the factor's structure +
 *    the working set's factor must fit 160 KB, otherwise the dense forms take the problem (qpc_qp_plan_info tells).
 *  - Streams: the *_device entry points enqueue on the caller's stream.  All launches of one
 *    controller on one device share a work counter, so they are serialised: when consecutive calls
 *    use DIFFERENT streams the library inserts an event dependency between them (no overlap, no
 *    corruption).  Under hipStreamBeginCapture the dependency cannot be recorded: capture every
 *    launch of a controller into ONE stream / graph and do not run the same controller on another
 *    stream while that graph executes.
 *  - Environment.  The product library reads: QPC_DISABLE_JIT=1 (use the ahead-of-time generic kernel instead of the
 *    hiprtc-specialised one), QPC_REQUIRE_JIT=1 (fail qpc_finalize if the specialisation cannot be compiled), QPC_NO_PRESOLVE=1
 *    (keep the hard foot-task rows in the QP instead of eliminating them), QPC_JIT_CACHE=0 (do not reuse compiled code objects
 *    within the process), QPC_JIT_CACHE_DIR=<directory> (keep them on disk as well: a later process with the same controller and
 *    library skips the ~5 s of hiprtc in qpc_finalize), QPC_HOST_SLOTS / QPC_HOST_CHUNK / QPC_ZERO_COPY (defaults of qpc_set_host_pipeline), and the
 *    diagnostics QPC_JIT_STAMPS=1 (in-kernel section timers) and QPC_DUMP_LDS=1 (print the LDS overlay map).  The A/B switches
 *    of the development history only exist in builds with -DQPC_EXPERIMENTS (libqpcontrol_hip_exp.so).
 *  - There is NO CPU fallback: every compute entry point needs a HIP device and fails with
 *    QPC_ERR_NO_DEVICE otherwise.
 */
#ifndef QPC_H
#define QPC_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define QPC_WORLD (-1)
#define QPC_FRAME_DEFAULT (-2) /* task frame = default frame of the task's target body */

/* error codes (return values) */
enum {
  QPC_OK = 0,
  QPC_ERR_INVALID_ARGUMENT = 1,
  QPC_ERR_NOT_FINALIZED = 2,
  QPC_ERR_ALREADY_FINALIZED = 3, /* reference: tasks/contacts must be added before the first tick,
                                    momentum.jl:43-46,150-156,182-187 */
  QPC_ERR_NO_DEVICE = 4,
  QPC_ERR_HIP = 5,
  QPC_ERR_UNSUPPORTED = 6, /* configuration exceeds the limits of the compiled kernels */
  QPC_ERR_OUT_OF_MEMORY = 7
};

/* joint types -- RigidBodyDynamics Revolute / Prismatic / QuaternionFloating / Fixed */
enum {
  QPC_JOINT_REVOLUTE = 0,      /* nq = nv = 1 */
  QPC_JOINT_PRISMATIC = 1,     /* nq = nv = 1 */
  QPC_JOINT_QUAT_FLOATING = 2, /* q = [w x y z px py pz], v = [omega_body; v_body], nq 7, nv 6 */
  QPC_JOINT_FIXED = 3          /* nq = nv = 0 */
};

/* motion task kinds -- the 7 task types of reference src/tasks.jl */
enum {
  QPC_TASK_SPATIAL_ACCELERATION = 0, /* src/tasks.jl:3-44    dim 6 */
  QPC_TASK_ANGULAR_ACCELERATION = 1, /* src/tasks.jl:47-84   dim 3 */
  QPC_TASK_LINEAR_ACCELERATION = 2,  /* src/tasks.jl:86-123  dim 3 */
  QPC_TASK_POINT_ACCELERATION = 3,   /* src/tasks.jl:125-171 dim 3 */
  QPC_TASK_JOINT_ACCELERATION = 4,   /* src/tasks.jl:173-189 dim nv(joint) */
  QPC_TASK_MOMENTUM_RATE = 5,        /* src/tasks.jl:192-236 dim 6 */
  QPC_TASK_LINEAR_MOMENTUM_RATE = 6  /* src/tasks.jl:239-262 dim 3 */
};

/* how a task enters the QP -- the three addtask! methods, momentum.jl:99-117 */
enum {
  QPC_WEIGHT_HARD = 0,          /* addtask!(c, task):      err == 0                      */
  QPC_WEIGHT_SCALAR = 1,        /* addtask!(c, task, w):   obj += w * (e . e)            */
  QPC_WEIGHT_MATRIX = 2,        /* addtask!(c, task, W):   obj += e' W e  (W dim x dim, row-major) */
  QPC_WEIGHT_SCALAR_PER_QP = 3, /* w is a Parameter: one scalar per QP in task_weights[]   */
  QPC_WEIGHT_MATRIX_PER_QP = 4  /* W is a Parameter: dim*dim per QP in task_weights[]      */
};

/* per-QP solve status -- mirrors checkstatus / QPSolveFailure, momentum.jl:83-91, exceptions.jl:1-11 */
enum {
  QPC_STATUS_OPTIMAL = 0,        /* MOI.OPTIMAL + FEASIBLE_POINT                      */
  QPC_STATUS_ALMOST_OPTIMAL = 1, /* MOI.ALMOST_OPTIMAL (tolerated by checkstatus)     */
  QPC_STATUS_INFEASIBLE = 2,
  QPC_STATUS_DUAL_INFEASIBLE = 3,
  QPC_STATUS_ITERATION_LIMIT = 4,
  QPC_STATUS_NUMERICAL_ERROR = 5 /* also: NaN / Inf anywhere in this QP's record (q, v, desired, contact weights and caps) */
};
/* A bad record never affects the other QPs of the batch and never keeps the kernel from terminating
 * (tests/test_bad_inputs.py).  Degenerate but legal records solve: zero contact weights, a zero cap (= disable!,
 * contacts.jl:72).  A negative cap is INFEASIBLE (0 <= rho <= rho_max < 0, contacts.jl:57); a negative weight makes the QP
 * non-convex (typically INFEASIBLE). */

typedef struct qpc_model qpc_model;
typedef struct qpc_controller qpc_controller;

/*
 * Flat description of a tree mechanism == RigidBodyDynamics `Mechanism{Float64}` after
 * remove_fixed_tree_joints! (reference notebooks/Standing controller.ipynb cell 2).
 * The transform of joint i maps body-i coordinates to parent coordinates:
 *     x_parent = joint_R[i] * (X_joint(q_i) * x_body) + joint_p[i]
 */
typedef struct qpc_model_desc {
  int32_t nbodies;
  const int32_t *parent;     /* [nbodies]   parent body or QPC_WORLD; parent[i] < i          */
  const int32_t *joint_type; /* [nbodies]   QPC_JOINT_*                                       */
  const double *joint_axis;  /* [nbodies*3] unit axis in the joint frame (revolute/prismatic) */
  const double *joint_R;     /* [nbodies*9] row-major rotation joint-frame-before -> parent body frame */
  const double *joint_p;     /* [nbodies*3] translation of the same transform                 */
  const double *mass;        /* [nbodies]                                                     */
  const double *com;         /* [nbodies*3] centre of mass in the body frame                  */
  const double *inertia;     /* [nbodies*6] Ixx Iyy Izz Ixy Ixz Iyz about the CoM, body axes  */
  double gravity[3];         /* gravitational acceleration in world frame, e.g. (0,0,-9.81)   */
} qpc_model_desc;

/* sizes of the frozen problem, reported by qpc_finalize / qpc_get_layout */
typedef struct qpc_layout {
  int32_t nq, nv;
  int32_t n_tasks, n_contacts, n_basis;
  int32_t n_desired;      /* sum of task dimensions: length of one QP's `desired` record        */
  int32_t n_task_weights; /* doubles per QP in `task_weights` (0 if no *_PER_QP weights)        */
  int32_t n_surface;      /* doubles per QP in `surfaces`: 4 per contact with a per-QP surface + 3 per contact with a per-QP position */
  /* the reference-layout QP this replaces, for bookkeeping: */
  int32_t ref_n_var, ref_n_eq, ref_n_ineq;
  /* the reduced QP the kernels actually solve: */
  int32_t n_free, n_pinned, n_slack, n_rho, n_eq_rows;
  int32_t lds_bytes_per_qp;
  int32_t n_desired_inputs; /* entries of `desired` still read from the caller (tasks without a
                             * desired law, qpc_task_set_law); 0 => `desired` may be NULL          */
} qpc_layout;

/* ---- model: mirrors Mechanism{Float64} ------------------------------------------------ */
int32_t qpc_model_create(const qpc_model_desc *desc, qpc_model **out);
void qpc_model_destroy(qpc_model *model);
int32_t qpc_model_nq(const qpc_model *model);
int32_t qpc_model_nv(const qpc_model *model);
/* first configuration / velocity index and counts of joint `body` (velocity_range, RBD) */
int32_t qpc_model_joint_ranges(const qpc_model *model, int32_t body, int32_t *q0, int32_t *nq,
                               int32_t *v0, int32_t *nv);

/* ---- controller: mirrors MomentumBasedController{N}(mechanism, optimizer; floatingjoint),
 *      momentum.jl:15-33.  floating_body = successor body of the floating joint or QPC_WORLD(-1)
 *      for `floatingjoint = nothing`.  n_basis == N (friction-cone basis size, contacts.jl:16). */
int32_t qpc_controller_create(const qpc_model *model, int32_t n_basis, int32_t floating_body,
                              qpc_controller **out);
void qpc_controller_destroy(qpc_controller *ctrl);

/* addtask! (momentum.jl:99-117) fused with the task constructor (src/tasks.jl:8-18,52-62,91-101,
 * 131-142,177-179,196-201,243-248).
 *   base_body/target_body: source/target of the kinematic path (ignored for JOINT/MOMENTUM kinds)
 *   frame_body: body whose frame the task is expressed in (SPATIAL/ANGULAR/LINEAR only;
 *               QPC_FRAME_DEFAULT = target body, QPC_WORLD = world). POINT tasks always use the
 *               base body's frame (src/tasks.jl:139-140,154).
 *   point: POINT task: point fixed in target body, target-body coordinates (src/tasks.jl:136-141)
 *   joint_body: JOINT task: the joint
 *   weight: NULL for HARD / *_PER_QP; 1 double for SCALAR; dim*dim row-major for MATRIX        */
int32_t qpc_add_task(qpc_controller *ctrl, int32_t kind, int32_t base_body, int32_t target_body,
                     int32_t frame_body, const double *point, int32_t joint_body,
                     int32_t weight_kind, const double *weight, int32_t *out_task_id);

/* Desired-acceleration LAWS evaluated on the device from the state of each QP -- the PD laws of
 * (controller::StandingController)(tau, t, state), src/highlevel/standing.jl:58-89, which in the
 * reference run on the CPU and call setdesired! before every tick.  A task with a law ignores
 * its slice of `desired`; when every task has one, `desired` may be NULL and the tick is a pure
 * (q, v) -> tau map.  gains = PDGains(kp, kd) (standing.jl:26-28).
 *   QPC_LAW_INPUT          (default) desired comes from the caller (setdesired!, src/tasks.jl:23..253)
 *   QPC_LAW_CONSTANT       desired = ref[0..dim)            (e.g. foot tasks: zero, standing.jl:37-38)
 *   QPC_LAW_JOINT_PD       JOINT task of a 1-DoF joint: kp (ref[0] - q) - kd v      (standing.jl:80-86)
 *   QPC_LAW_COM_PD         LINEAR_MOMENTUM_RATE task: m (-kp (c - ref[0..3)) - kd h_lin / m), c = centre
 *                          of mass, h = momentum(state)                              (standing.jl:60-67)
 *   QPC_LAW_ORIENTATION_PD ANGULAR task with base = world, frame = target body: -kp rotation_vector(R)
 *                          - kd R' omega  of the target body                          (standing.jl:70-77)
 *   QPC_LAW_SE3_PD         SPATIAL task expressed in its target body's frame: the reference's SE3PDController
 *                          (src/lowlevel/se3pdcontroller.jl:13-18) with its SE3Trajectory (src/trajectories/se3.jl)
 *                          evaluated on the device:  desired = Tdref + pd(gains, H, Href, T, Tref), where
 *                          (Href, Tref, Tdref) = trajectory(t, Val(2)) and H, T are the pose / twist of the body w.r.t.
 *                          the base of THIS QP's state.  kp, kd = angular gains, ref[0..1] = linear gains (kp, kd).
 *                          The task's slice of `desired` becomes a per-QP TRAJECTORY RECORD of QPC_SE3_RECORD doubles
 *                          (qpc_task_desired_range tells where): [0..5] unused, [6] t, [7] x0, [8] xf, [9..14]
 *                          c0..c5 of the interpolator alpha(theta) = sum c_i theta^i (identity: 0,1,0,0,0,0; a
 *                          fit_quintic warp: its coefficients), [15..18] quaternion y0 (w x y z), [19..22] quaternion
 *                          yf (rotation body -> base at x0 / xf, interpolated.jl:68-83), [23..25] p0, [26..28] pf
 *                          (translation body -> base).  theta = clamp((t - x0) / (xf - x0), 0, 1) (interpolated.jl:29-40).
 *                          PIECEWISE trajectories (src/trajectories/piecewise.jl:1-30: `Piecewise(subfunctions, breaks)`, clamped): pass
 *                          ref[2] = number of pieces np (2 .. QPC_SE3_MAX_PIECES; n_ref = 3).  The record then has
 *                          QPC_SE3_PW_RECORD(np) = 8 + 23 np doubles: [0..5] unused, [6] t, [7 .. 7+np] the np + 1 breaks, then per
 *                          piece the 22 doubles [x0 xf | c0..c5 | y0 | yf | p0 | pf] of an Interpolated pair as above, in the piece's
 *                          LOCAL time (piecewise.jl:25: the piece is evaluated at clamp(t) - breaks[index]); the piece is chosen
 *                          per QP on the device (index = min(searchsortedlast(breaks, clamp(t)), np)). */
enum {
  QPC_LAW_INPUT = 0,
  QPC_LAW_CONSTANT = 1,
  QPC_LAW_JOINT_PD = 2,
  QPC_LAW_COM_PD = 3,
  QPC_LAW_ORIENTATION_PD = 4,
  QPC_LAW_SE3_PD = 5
};
#define QPC_SE3_RECORD 29
#define QPC_SE3_MAX_PIECES 4
#define QPC_SE3_PW_RECORD(np) (8 + 23 * (np))
int32_t qpc_task_set_law(qpc_controller *ctrl, int32_t task_id, int32_t law, double kp, double kd,
                         const double *ref, int32_t n_ref);
/* where task `task_id` lives in one QP's `desired` record (after qpc_finalize): first entry and length
 * (the task dimension, or QPC_SE3_RECORD for a task with the SE3 PD law) */
int32_t qpc_task_desired_range(const qpc_controller *ctrl, int32_t task_id, int32_t *offset, int32_t *length);

/* regularize!(c, joint, w): obj += w * |vdot_joint|^2 ; weights add up (momentum.jl:128-131) */
int32_t qpc_regularize(qpc_controller *ctrl, int32_t joint_body, double weight);
/* regularize!(c, joint, weight) with a Parametron `Parameter` as the weight (momentum.jl:128-131 takes any `weight`; the
 * Parameter is re-evaluated every tick): the weight is a per-QP input, entry *out_weight_index of `task_weights`.  At most one
 * per-QP weight per joint; static weights (qpc_regularize) still add to it.  Like every per-QP task weight it switches the
 * static-Hessian fast path of the tick off. */
int32_t qpc_regularize_per_qp(qpc_controller *ctrl, int32_t joint_body, int32_t *out_weight_index);

/* addcontact!(c, body, position, normal, mu) (momentum.jl:133-148, contacts.jl:38-69).
 * position/normal in the frame of `body`.  The reference lets position, normal and mu be Parameters
 * (`position::Union{Point3D, Parameter{<:Point3D}}` ..., contacts.jl:39); here `per_qp_surface` is a
 * bit set: QPC_CONTACT_SURFACE_PER_QP (1): normal and mu come per QP as [nx ny nz mu] in `surfaces`;
 * QPC_CONTACT_POSITION_PER_QP (2): the position comes per QP as [px py pz].  A contact's record in
 * `surfaces` is [nx ny nz mu] (if bit 1) followed by [px py pz] (if bit 2); the records of all such
 * contacts are concatenated in contact order (layout.n_surface doubles per QP).  The static arguments
 * are ignored for what is per QP.  (The closed-loop step qpc_sim_step_batch feels the ground at the
 * STATIC positions.)  A fresh contact has weight = 0 and maxnormalforce = 0 in the reference
 * (contacts.jl:50); here both are per-QP inputs of qpc_control_batch. */
enum { QPC_CONTACT_SURFACE_PER_QP = 1, QPC_CONTACT_POSITION_PER_QP = 2 };
int32_t qpc_add_contact(qpc_controller *ctrl, int32_t body, const double position[3],
                        const double normal[3], double mu, int32_t per_qp_surface,
                        int32_t *out_contact_id);

/* initialize! (momentum.jl:150-156): freeze the layout, add the wrench-balance rows when a
 * floating joint was given (momentum.jl:162-193), compile the task program and upload it to
 * HIP device `device_id`.  What the compile step decides once (qpc_compile.cpp): hard JointAccelerationTasks
 * and hard body-frame tasks of the floating body become pinned accelerations; hard 6-row SpatialAccelerationTasks whose path
 * holds six free one-DoF joints (the foot tasks of a StandingController, registered first) are eliminated per QP by a
 * null-space presolve, and the soft task rows that remain are folded into the objective -- the same QP, solved in fewer
 * dimensions (layout.n_free / n_eq_rows report the problem as assembled). */
int32_t qpc_finalize(qpc_controller *ctrl, int32_t device_id, qpc_layout *out_layout);
/* Same for SEVERAL devices of this process: the task program is replicated on every
 * device of device_ids[0..n_devices) and the host-pointer entry points (qpc_control_batch,
 * qpc_control_batch_async, qpc_rollout_batch) shard every batch into contiguous ranges, one per
 * device, QP index range [g B/G, (g+1) B/G) -> device_ids[g]; no collective, the QPs are independent.
 * The *_device entry points address device_ids[0] unless they take a device index. */
int32_t qpc_finalize_devices(qpc_controller *ctrl, int32_t n_devices, const int32_t *device_ids,
                             qpc_layout *out_layout);
int32_t qpc_n_devices(const qpc_controller *ctrl);
int32_t qpc_get_layout(const qpc_controller *ctrl, qpc_layout *out_layout);

/* Host-pointer pipeline (before or after qpc_finalize*; not while a job is in flight): every device's
 * shard of a batch is cut into chunks of `chunk_qps` QPs (<= 0: keep; default 4096) that flow through
 * `slots_per_device` pipeline slots per device (<= 0: keep; default 6, at most 8; only before qpc_finalize*), one
 * worker thread each -- staging one chunk, the kernel of another and un-staging a third overlap.  (Measured on MI355X with
 * pageable arrays: 6 slots 16.7-17.0 M solves/s, 8: 13-16 M, 10: 4.5-5.6 M -- every slot's launch is a persistent grid that
 * fills the device, and beyond 8 they starve each other.)  Batches of at
 * most `inline_qps` QPs (< 0: keep; default 2048) are run on the caller's thread without a hand-off. */
int32_t qpc_set_host_pipeline(qpc_controller *ctrl, int64_t chunk_qps, int32_t slots_per_device,
                              int64_t inline_qps);

/* Page-locked host memory.  Arrays handed to the host-pointer entry points that live in pinned memory
 * (from qpc_host_alloc, qpc_host_register, hipHostMalloc / hipHostRegister or torch pin_memory) are
 * used IN PLACE: when every array of a call is page-locked the kernels read the records and write
 * the results over PCIe directly (one launch per device, no staging, no chunks; measured 89 % of the
 * device-resident rate).  Pageable arrays are staged through the library's own pinned buffers (one
 * extra host copy each way, done chunk by chunk by the pipeline's worker threads). */
void *qpc_host_alloc(uint64_t bytes);
void qpc_host_free(void *ptr);
int32_t qpc_host_register(void *ptr, uint64_t bytes);
int32_t qpc_host_unregister(void *ptr);

/* solver knobs (all optional; defaults in parentheses) */
typedef struct qpc_settings {
  int32_t max_active_set_iters; /* (200) -> QPC_STATUS_ITERATION_LIMIT                      */
  int32_t max_refine_iters;     /* (6) proximal/iterative-refinement sweeps                */
  double eps_primal_reg;        /* (1e-9) relative primal proximal regularisation          */
  double eps_dual_reg;          /* (1e-11) relative dual regularisation of equality rows   */
  double tol_residual;          /* (1e-10) relative KKT residual for OPTIMAL               */
  double tol_bound;             /* (1e-10) relative bound violation / multiplier sign tol  */
} qpc_settings;
int32_t qpc_default_settings(qpc_settings *s);
/* Arithmetic of the tick kernel (before qpc_finalize).  QPC_PRECISION_F64 (default): everything in fp64, tau within 1e-8 of the
 * CPU OSQP path (north star).  QPC_PRECISION_F32 -- BASELINE config 5's "fp32" variant, modelled on the mixed-contact-set test
 * /root/reference/test/controller.jl:169-230: the SAME device source with `real` = float (LDS slab half the size, packed fp32
 * arithmetic where the compiler finds pairs, v_mfma_f32_16x16x4_f32 for the Schur build), every tolerance of the solver moved
 * to the fp32 rounding floor (regularisation 1e-5, pivots 1e-5, bounds / acceptance 1e-4 ...; the qpc_settings tolerances are
 * ignored).  Inputs and outputs stay fp64 arrays (converted at the kernel's boundary), so the call is a drop-in.  Accuracy against
 * the fp64 path, relative to the torque scale (measured on 32 768 Atlas states per config): median 6e-7, 99.9 % below 5e-5
 * (SURVEY.md 8d row 5 asks 1e-4), a handful of QPs in 10^4 up to 8e-3.  What fp32 cannot resolve is reported, not guessed: a
 * floating robot on ONE enabled contact point -> NUMERICAL_ERROR; a search that still cycles after 48 working-set changes (3 QPs
 * in 10^4) -> ITERATION_LIMIT; a primal residual within 10x of the tolerance -> NUMERICAL_ERROR instead of INFEASIBLE.
 * Needs the hiprtc-specialised kernel (QPC_ERR_UNSUPPORTED with QPC_DISABLE_JIT=1); the closed-loop entry points
 * (qpc_sim_step_batch, qpc_rollout_batch) and warm starts stay fp64-only. */
/* Lane packing of the tick kernel (before qpc_finalize).  1 (default): one QP per wavefront.  2: two QPs per wavefront -- lanes 0-31 and
 * 32-63 each work their own QP on their own LDS slab, one instruction stream advances both (the tick uses at most ~36 of its 64
 * lanes at a time and is bound by the number of instructions it issues; DESIGN.md 7c has the measurements).  Same results (the
 * per-QP arithmetic does not change); needs the hiprtc-specialised kernel, a contact basis of four rays and at most two presolved
 * foot blocks and no warm start (qpc_finalize refuses the combination with qpc_enable_warm_start); the closed-loop entry points
 * (qpc_rollout_batch*) run the packed tick. */
int32_t qpc_set_qps_per_wavefront(qpc_controller *ctrl, int32_t qps_per_wavefront);
enum { QPC_PRECISION_F64 = 0, QPC_PRECISION_F32 = 1 };
int32_t qpc_set_precision(qpc_controller *ctrl, int32_t precision);
int32_t qpc_set_settings(qpc_controller *ctrl, const qpc_settings *s);

/* L2 -- the graded path: the callable (controller)(tau, t, x), momentum.jl:41-81, with
 * setdesired! (src/tasks.jl:23,67,106,147,183,225,253), point.weight[] and
 * point.maxnormalforce[] (contacts.jl:35-36) folded into per-QP input arrays.
 *   in : q[nq*B], v[nv*B], desired[n_desired*B], contact_weight[nc*B],
 *        contact_maxnormalforce[nc*B], task_weights[n_task_weights*B] or NULL,
 *        surfaces[n_surface*B] or NULL
 *   out: tau[nv*B], vdot[nv*B] (result.vdot, momentum.jl:62-64),
 *        contact_wrench[6*nc*B] (value(wrench_world) per contact, world frame, [ang;lin],
 *        contacts.jl:82-89) or NULL, status[B] (QPC_STATUS_*), iters[B] or NULL (diagnostic: working-set changes of the
 *        search + 1000 x refinement sweeps of the final solve + 1000000 if the QP was solved in the reduced dimensions of
 *        the null-space presolve)
 * Host-pointer variant: synchronous; the batch is sharded over the controller's devices and
 * pipelined in chunks (H2D || kernel || D2H, qpc_set_host_pipeline). */
int32_t qpc_control_batch(qpc_controller *ctrl, int64_t B, const double *q, const double *v,
                          const double *desired, const double *contact_weight,
                          const double *contact_maxnormalforce, const double *task_weights,
                          const double *surfaces, double *tau, double *vdot,
                          double *contact_wrench, int32_t *status, int32_t *iters);
/* Same, but returns as soon as the job is posted to the controller's worker threads; every array
 * must stay valid and untouched until qpc_wait(ctrl) returns.  One job in flight per controller: a
 * second *_async call first waits for the previous job.  qpc_wait returns the error of the job
 * (QPC_OK if there is none in flight).  The error of an async job is reported by qpc_wait ONLY: any other
 * host-pointer call on the controller (synchronous tick, another *_async, rollout, L1) first lets the job in
 * flight finish, drops its error if nobody waited for it, and then runs and reports its OWN batch -- a
 * synchronous call never returns an error for work it did not attempt. */
int32_t qpc_control_batch_async(qpc_controller *ctrl, int64_t B, const double *q, const double *v,
                                const double *desired, const double *contact_weight,
                                const double *contact_maxnormalforce, const double *task_weights,
                                const double *surfaces, double *tau, double *vdot,
                                double *contact_wrench, int32_t *status, int32_t *iters);
int32_t qpc_wait(qpc_controller *ctrl);
/* The error such a dropped job had (QPC_OK if none): the code is returned, the text copied to `message` (may be NULL), and
 * the record is cleared.  A dropped failure is also announced once on stderr; the outputs of that batch are not valid. */
int32_t qpc_last_async_error(qpc_controller *ctrl, char *message, int32_t message_bytes);

/* Same, but every pointer is a DEVICE pointer on the controller's device and the work is
 * enqueued on `hip_stream` (a hipStream_t, NULL = default stream) without synchronising. */
int32_t qpc_control_batch_device(qpc_controller *ctrl, int64_t B, const double *q,
                                 const double *v, const double *desired,
                                 const double *contact_weight,
                                 const double *contact_maxnormalforce,
                                 const double *task_weights, const double *surfaces, double *tau,
                                 double *vdot, double *contact_wrench, int32_t *status,
                                 int32_t *iters, void *hip_stream);
/* ---- cost-ordered tickets.  A launch hands its QPs out in batch order and drains over the duration of its longest working-set
 *      search (~110 us: 9 % of a 32 768-QP launch, a quarter of an 8 192-QP one).  Nothing can know a QP's cost before it is
 *      solved -- except a control loop, which solves (nearly) the same model every tick (momentum.jl:58: solve! on the same model):
 *      the previous tick's `iters` of robot b predicts this tick's.  qpc_ticket_order_device sorts the QP indices by descending
 *      key (key = cost[b] % modulus if modulus > 0 else cost[b], clamped to 0 .. 255; pass the `iters` output with modulus 1000,
 *      or any static predictor such as the number of enabled contacts) into order[B] (device memory, on `hip_stream`);
 *      qpc_control_batch_device_ordered then starts the expensive QPs first.  Outputs are bit-identical to the unordered call
 *      (every QP is independent); order == NULL is the unordered call. */
int32_t qpc_ticket_order_device(qpc_controller *ctrl, int32_t device_index, int64_t B, const int32_t *cost, int32_t modulus,
                                int32_t *order, void *hip_stream);
int32_t qpc_control_batch_device_ordered(qpc_controller *ctrl, int32_t device_index, int32_t slot, int64_t B, const double *q,
                                         const double *v, const double *desired, const double *contact_weight,
                                         const double *contact_maxnormalforce, const double *task_weights,
                                         const double *surfaces, double *tau, double *vdot, double *contact_wrench,
                                         int32_t *status, int32_t *iters, const int32_t *order, void *hip_stream);
/* Same on device device_ids[device_index] of a multi-device controller (qpc_finalize_devices). */
int32_t qpc_control_batch_device_on(qpc_controller *ctrl, int32_t device_index, int64_t B,
                                    const double *q, const double *v, const double *desired,
                                    const double *contact_weight,
                                    const double *contact_maxnormalforce,
                                    const double *task_weights, const double *surfaces,
                                    double *tau, double *vdot, double *contact_wrench,
                                    int32_t *status, int32_t *iters, void *hip_stream);

/* Same on launch slot `slot` (0 .. QPC_USER_SLOTS - 1) of that device.  Every slot has its own work counter, so launches of
 * DIFFERENT slots on different streams may overlap; launches of one slot are serialised (a change of stream becomes an event
 * dependency).  The entry points above use slot 0.  A persistent launch drains over the duration of its longest QP (about 110 us
 * for the Atlas controller): with two independent batches in flight on two slots the workgroups of one fill the tail of the
 * other (bench.py config.pipelined).  The batches must use distinct output arrays. */
#define QPC_USER_SLOTS 4
int32_t qpc_control_batch_device_slot(qpc_controller *ctrl, int32_t device_index, int32_t slot, int64_t B,
                                      const double *q, const double *v, const double *desired,
                                      const double *contact_weight,
                                      const double *contact_maxnormalforce,
                                      const double *task_weights, const double *surfaces,
                                      double *tau, double *vdot, double *contact_wrench,
                                      int32_t *status, int32_t *iters, void *hip_stream);

/* Warm start -- the counterpart of the reference's persistent OSQP workspace (momentum.jl:58 calls
 * solve! on the SAME model every tick, so OSQP starts from the previous tick's iterate).  Here the
 * carried state is the working set of the active-set search: per QP two 64-bit masks,
 * working_set[2 b] bit j = rho_j at its lower bound (0), working_set[2 b + 1] bit j = rho_j at its
 * upper bound (maxrho, contacts.jl:57), j = N * contact + ray.  The caller keeps the array between
 * ticks (zeros = cold start); the tick reads it as the starting set and overwrites it with the final
 * one.  The result never depends on the hint: a search that fails from a warm set is repeated from the
 * cold start inside the kernel.  qpc_enable_warm_start(ctrl, 1) before qpc_finalize builds the
 * (separately compiled) warm-started kernel next to the cold one, whose speed it does not touch; it
 * needs the hiprtc-specialised kernel and at most 64 rho's (else QPC_ERR_UNSUPPORTED at qpc_finalize).
 * qpc_rollout_batch[_device] then warm-start their ticks by themselves. */
int32_t qpc_enable_warm_start(qpc_controller *ctrl, int32_t enable);
int32_t qpc_control_batch_warm(qpc_controller *ctrl, int64_t B, const double *q, const double *v,
                               const double *desired, const double *contact_weight,
                               const double *contact_maxnormalforce, const double *task_weights,
                               const double *surfaces, double *tau, double *vdot,
                               double *contact_wrench, int32_t *status, int32_t *iters,
                               uint64_t *working_set);
int32_t qpc_control_batch_device_warm(qpc_controller *ctrl, int32_t device_index, int64_t B,
                                      const double *q, const double *v, const double *desired,
                                      const double *contact_weight,
                                      const double *contact_maxnormalforce,
                                      const double *task_weights, const double *surfaces,
                                      double *tau, double *vdot, double *contact_wrench,
                                      int32_t *status, int32_t *iters, uint64_t *working_set,
                                      void *hip_stream);

/* L1 -- assembly + solve from HOST-EVALUATED blocks: `solve!` + `value` of the reference
 * (momentum.jl:58-72) with the kinematics taken out.  Per QP the caller supplies
 * what the reference evaluates from the MechanismState:
 *   task_J[n_desired*nv*B], task_bias[n_desired*B]  rows of all tasks stacked in task order,
 *        task_error = J vdot + bias - desired   (src/tasks.jl:31-44,73-84,112-123,153-171,185-189,230-236,258-262)
 *   momentum_A[6*nv*B], momentum_rhs[6*B]           world-frame momentum matrix and W_g - Adot v of the
 *        wrench balance (momentum.jl:162-193); NULL without a floating joint
 *   contact_R[9*nc*B], contact_p[3*nc*B]            world pose of each z-up contact frame (contacts.jl:61)
 *   desired, contact_weight, contact_maxnormalforce, task_weights as in qpc_control_batch;
 *   surfaces: only mu ([.. .. .. mu] per per-QP contact) is read, the normal is already in contact_R
 * Out: vdot[nv*B], contact_wrench[6*nc*B] (may be NULL), status[B], iters[B] (may be NULL).
 * Dense one-lane-per-QP path sharing nothing with the fused tick kernel: parity / cross-check, not
 * the performance path.  Host pointers, synchronous; the controller must be finalized. */
int32_t qpc_assemble_solve_batch(qpc_controller *ctrl, int64_t B, const double *task_J,
                                 const double *task_bias, const double *desired,
                                 const double *momentum_A, const double *momentum_rhs,
                                 const double *contact_R, const double *contact_p,
                                 const double *contact_weight, const double *contact_maxnormalforce,
                                 const double *task_weights, const double *surfaces, double *vdot,
                                 double *contact_wrench, int32_t *status, int32_t *iters);

/* L0 -- raw dense QPs in OSQP problem form (what Parametron.solve! hands to the MOI optimizer,
 * momentum.jl:58; settings as test/runtests.jl:35-43):
 *     min 1/2 x'Px + q'x   s.t.  l <= A x <= u
 * P[n*n*B] (symmetric, row-major per QP), q[n*B], A[m*n*B] row-major, l[m*B], u[m*B]
 * (+-1e20 or inf = unbounded); out x[n*B], y[m*B] (multipliers, OSQP sign convention),
 * status[B], iters[B].  Host pointers, synchronous. */
int32_t qpc_qp_solve_batch(int32_t n, int32_t m, int64_t B, const double *P, const double *q,
                           const double *A, const double *l, const double *u,
                           const qpc_settings *settings, int32_t device_id, double *x, double *y,
                           int32_t *status, int32_t *iters);
/* Same with DEVICE pointers on device `device_id`, enqueued on `hip_stream` without synchronising;
 * `workspace` = at least qpc_qp_workspace_bytes(n, m, B) bytes of device memory owned by the caller
 * (the host-pointer variant keeps a persistent one per device). */
uint64_t qpc_qp_workspace_bytes(int32_t n, int32_t m, int64_t B);
int32_t qpc_qp_solve_batch_device(int32_t n, int32_t m, int64_t B, const double *P, const double *q,
                                  const double *A, const double *l, const double *u,
                                  const qpc_settings *settings, int32_t device_id, double *x,
                                  double *y, int32_t *status, int32_t *iters, void *workspace,
                                  uint64_t workspace_bytes, void *hip_stream);

/* L0 with the OPTIMIZER'S LIFE CYCLE -- what `QPCOptimizer <: MOI.AbstractOptimizer` (julia/QPControlHIPMOI.jl) binds.  Parametron
 * hands the model to the optimizer once (`MOI.copy_to`: a fixed sparsity pattern) and before every `MOI.optimize!` rewrites the
 * coefficients in place (momentum.jl:58 `solve!`); OSQP, the reference's optimizer (test/runtests.jl:35-43), splits the work the same
 * way (osqp_setup / osqp_update_* / osqp_solve, warm-started by the previous solution).
 *   qpc_qp_setup: the PATTERN, 0-based COO, entries unique.  P: a symmetric matrix, an entry (i, j), i != j, stands for P_ij = P_ji
 *     (give one triangle, as OSQP does).  A: m x n.  What depends on the pattern alone is computed here, once, and kept on device
 *     `device_id`.
 *   qpc_qp_solve: B QPs with that pattern: P_vals[nnz_P*B], A_vals[nnz_A*B] in the pattern's order (QP-major), q[n*B], l[m*B],
 *     u[m*B]; out as qpc_qp_solve_batch.  warm_start != 0: the working set starts from the multipliers of the previous call (QP b
 *     from QP b; ignored on the first call or when the previous batch was smaller).  Host pointers, synchronous; one call at a
 *     time per problem.  13 KB per QP cross the bus for the 143 x 178 Atlas problem (dense entry: 368 KB).
 *   Three solvers stand behind qpc_qp_solve, each taking what the one before leaves (all gfx950 device code):
 *     1. the SPARSE workgroup form (csrc/qpc_qp_sp.h): n, m <= 256 and P block diagonal with blocks of at most 8 variables (diagonal P;
 *        the e'We of a matrix-weighted task, momentum.jl:113-117).  It follows a PLAN: the symbolic analysis of G = A H^-1 A' for
 *        one set of equality rows (rows with l == u) -- fill-reducing order, elimination tree, fronts, update schedule
 *        (csrc/qpc_qp_sym.h) -- made on the host at the first qpc_qp_solve from the batch's first QP (a few milliseconds), or
 *        ahead of it by qpc_qp_analyze, and redone when a batch's first QP has other equality rows.  A QP of a batch whose
 *        equality rows differ from the plan's is left to
 *     2. the dense workgroup form (csrc/qpc_qp_wg.h): diagonal P, n, m <= 256; and
 *     3. one lane per QP (csrc/qpc_qp_dense.h): anything -- a dense P among it -- at 0.6 s per Atlas-sized problem.
 *   qpc_qp_analyze: makes the plan now, for the equality rows of one QP's bounds l[m], u[m] (`MOI.copy_to` knows the constraint
 *     sets: an optimizer calls it once after qpc_qp_setup so that the first tick is not the slow one).  QPC_OK also when the
 *     pattern is outside the sparse form (qpc_qp_plan_info tells).
 *   qpc_qp_plan_info: out[0 .. n_out-1] <- { plan in use (0 / 1), equality rows, other rows, fronts, levels of the elimination
 *     tree, update jobs, doubles of the value store, bytes of LDS per workgroup, widest front, tallest front, multiply-adds of the
 *     elimination, multiply-adds of the dense elimination of the same rows }; missing trailing entries are not written. */
typedef struct qpc_qp_problem qpc_qp_problem;
int32_t qpc_qp_setup(int32_t n, int32_t m, int64_t nnz_P, const int32_t *P_rows, const int32_t *P_cols,
                     int64_t nnz_A, const int32_t *A_rows, const int32_t *A_cols, int32_t device_id,
                     qpc_qp_problem **out);
int32_t qpc_qp_solve(qpc_qp_problem *problem, int64_t B, const double *P_vals, const double *q,
                     const double *A_vals, const double *l, const double *u,
                     const qpc_settings *settings, int32_t warm_start, double *x, double *y,
                     int32_t *status, int32_t *iters);
int32_t qpc_qp_analyze(qpc_qp_problem *problem, const double *l, const double *u);
/* qpc_qp_solve on DEVICE pointers (memory of the problem's device): values, bounds and results stay in HBM -- for a caller that builds
 * its QPs on the GPU, and the rate of the path without the 13 KB per problem over PCIe.  Same arguments, same three forms in the same
 * order, same results bit for bit.  The plan must exist (qpc_qp_analyze with one QP's bounds from the host, or an earlier
 * qpc_qp_solve): without one the dense forms take the whole batch.  Work is queued on `hip_stream` (0: the default stream); the call
 * returns after it has synchronised that stream -- once per form that ran, normally once: between the forms the host learns how many
 * QPs are left.  One call at a time per problem (the forms share the problem's counter and its warm-start multipliers). */
int32_t qpc_qp_solve_device(qpc_qp_problem *problem, int64_t B, const double *P_vals, const double *q,
                            const double *A_vals, const double *l, const double *u,
                            const qpc_settings *settings, int32_t warm_start, double *x, double *y,
                            int32_t *status, int32_t *iters, void *hip_stream);
int32_t qpc_qp_plan_info(const qpc_qp_problem *problem, int64_t *out, int32_t n_out);
/* Measurement: with `on` != 0 every qpc_qp_solve brackets its kernels with HIP events on the problem's own stream;
 * qpc_qp_last_kernel_ms: the device time of the last call's kernels (all three forms together), milliseconds.  bench.py's
 * `roofline_l0` is computed from it. */
int32_t qpc_qp_enable_timing(qpc_qp_problem *problem, int32_t on);
int32_t qpc_qp_last_kernel_ms(const qpc_qp_problem *problem, double *ms);
void qpc_qp_free(qpc_qp_problem *problem);

/* ---- closed loop: what the reference's acceptance test does around the tick,
 *      notebooks/Standing controller.ipynb cells 9-11 -- `Dynamics(mechanism, controller)` with a
 *      soft contact model, stepped by RigidBodySim `simulate` -- for a batch of robots on the device.
 * Ground: the half space z <= ground_z, felt at the controller's contact points (qpc_add_contact:
 * body, position, mu): normal force max(stiffness*penetration - damping*v_z, 0), tangential force
 * -tangential_damping*v_xy capped at mu * normal force.  Integrator: semi-implicit Euler,
 * v += dt*vdot, q <- q (+) v*dt on the configuration manifold (unit quaternion renormalised). */
typedef struct qpc_sim_params {
  double dt;                 /* step, e.g. 1e-3 (the control rate)  */
  double ground_z;           /* height of the ground plane          */
  double stiffness;          /* N/m   per contact point             */
  double damping;            /* N s/m per contact point, normal     */
  double tangential_damping; /* N s/m per contact point, tangential */
} qpc_sim_params;

/* One forward-dynamics step  M(q) vdot = tau - c(q, v, ground)  per robot, in place on q[nq*B], v[nv*B].
 * tau[nv*B]: joint torques (floating-joint entries are applied as given: the tick writes zeros there,
 * momentum.jl:93-97).  status[B] or NULL: tick statuses; a robot whose QP failed (status beyond
 * ALMOST_OPTIMAL, checkstatus momentum.jl:83-91) gets zero torque for the step.  bad[B] or NULL:
 * incremented for such a robot (and for a non-positive-definite mass matrix, which leaves the state
 * untouched).  Host pointers, synchronous; the controller must be finalized. */
int32_t qpc_sim_step_batch(qpc_controller *ctrl, int64_t B, double *q, double *v, const double *tau,
                           const int32_t *status, const qpc_sim_params *params, int32_t *bad);
/* Same with DEVICE pointers, enqueued on `hip_stream` without synchronising. */
int32_t qpc_sim_step_batch_device(qpc_controller *ctrl, int64_t B, double *q, double *v,
                                  const double *tau, const int32_t *status,
                                  const qpc_sim_params *params, int32_t *bad, void *hip_stream);

/* n_steps x (controller tick -> simulation step) without leaving the device: the batched counterpart
 * of `simulate(problem, ...)` in the notebook.  q, v are advanced in place; the tick inputs (desired,
 * contact_weight, ...) are as in qpc_control_batch and held constant over the rollout (`desired` may be
 * NULL when every task has a desired law, qpc_task_set_law).  Out: tau[nv*B] and status[B] of the LAST
 * tick (either may be NULL), bad[B] = number of steps whose QP failed per robot (may be NULL).
 * Host pointers, synchronous. */
int32_t qpc_rollout_batch(qpc_controller *ctrl, int64_t B, int64_t n_steps, double *q, double *v,
                          const double *desired, const double *contact_weight,
                          const double *contact_maxnormalforce, const double *task_weights,
                          const double *surfaces, const qpc_sim_params *params, double *tau,
                          int32_t *status, int32_t *bad);
/* Same with DEVICE pointers on `hip_stream`, not synchronised.  tau[nv*B], vdot[nv*B], status[B] are
 * caller-provided work/output buffers (all required); bad[B] may be NULL. */
int32_t qpc_rollout_batch_device(qpc_controller *ctrl, int64_t B, int64_t n_steps, double *q, double *v,
                                 const double *desired, const double *contact_weight,
                                 const double *contact_maxnormalforce, const double *task_weights,
                                 const double *surfaces, const qpc_sim_params *params, double *tau,
                                 double *vdot, int32_t *status, int32_t *bad, void *hip_stream);

/* QPSolveFailure text (exceptions.jl:7-11) */
const char *qpc_status_string(int32_t status);
/* The three fields of QPSolveFailure (exceptions.jl:1-5: terminationstatus, primalstatus, dualstatus -- what checkstatus,
 * momentum.jl:83-91, reads back from the optimizer) for a QPC_STATUS_* code, as the names of the MOI 0.8 enum values
 * (MOI.TerminationStatusCode / MOI.ResultStatusCode) the OSQP wrapper of the reference reports in the same situation:
 *   OPTIMAL          -> OPTIMAL,          FEASIBLE_POINT,            FEASIBLE_POINT
 *   ALMOST_OPTIMAL   -> ALMOST_OPTIMAL,   UNKNOWN_RESULT_STATUS,     UNKNOWN_RESULT_STATUS   (accepted by checkstatus)
 *   INFEASIBLE       -> INFEASIBLE,       NO_SOLUTION,               INFEASIBILITY_CERTIFICATE
 *   DUAL_INFEASIBLE  -> DUAL_INFEASIBLE,  INFEASIBILITY_CERTIFICATE, NO_SOLUTION
 *   ITERATION_LIMIT  -> ITERATION_LIMIT,  UNKNOWN_RESULT_STATUS,     UNKNOWN_RESULT_STATUS
 *   NUMERICAL_ERROR  -> NUMERICAL_ERROR,  UNKNOWN_RESULT_STATUS,     UNKNOWN_RESULT_STATUS
 * (the active-set solver proves infeasibility by an inconsistent equality block / an unbounded Goldfarb-Idnani step and
 * does not form a Farkas vector: the dual status names what OSQP would report, the certificate itself is not returned).
 * Any of the three output pointers may be NULL; returns QPC_ERR_INVALID_ARGUMENT for an unknown code. */
int32_t qpc_status_triple(int32_t status, const char **termination, const char **primal, const char **dual);

/* ---- `contactwrenches[BodyID(body)]` (momentum.jl:64-72): the reference hands inverse_dynamics! -- and leaves in its result -- ONE
 *      wrench per body that carries contact points, the SUM of its points' wrenches (world frame, about the world origin); the tick's
 *      `contact_wrench` output is per contact point (6 nc per QP).
 * qpc_contact_bodies: *n_bodies <- bodies with contact points; bodies[n] <- their ids in the order of their first
 *   qpc_add_contact (`keys(controller.contacts)`), body_of_contact[nc] <- index of each contact's body in that list.  Either array
 *   may be NULL.
 * qpc_contact_wrench_per_body: body_wrench[6 n_bodies * B] <- sums of contact_wrench[6 nc * B] (host arrays; `_device`: device arrays
 *   on the controller's device device_ids[device_index], enqueued on `hip_stream`). */
int32_t qpc_contact_bodies(const qpc_controller *ctrl, int32_t *n_bodies, int32_t *bodies, int32_t *body_of_contact);
int32_t qpc_contact_wrench_per_body(const qpc_controller *ctrl, int64_t B, const double *contact_wrench, double *body_wrench);
int32_t qpc_contact_wrench_per_body_device(qpc_controller *ctrl, int32_t device_index, int64_t B, const double *contact_wrench,
                                           double *body_wrench, void *hip_stream);
const char *qpc_last_error(void);
const char *qpc_version(void);
/* 16 hex digits identifying the device code of this build (hash of the kernel sources): stored with every
 * profile under profiles/ so that a quoted measurement can be matched to the kernel that produced it */
const char *qpc_kernel_fingerprint(void);

/* kernel timing hooks for bench.py / roofline: average duration in ms of the fused tick kernel
 * over launches recorded since the last reset, measured with hipEvents on the launch stream. */
int32_t qpc_timing_enable(qpc_controller *ctrl, int32_t enable);
int32_t qpc_timing_read(qpc_controller *ctrl, double *avg_ms, int64_t *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* QPC_H */
