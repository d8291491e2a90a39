# QPControlHIP.jl -- thin Julia host over libqpcontrol_hip.so (include/qpc.h).
#
# Mirrors the exported surface of QPControl.jl (reference src/QPControl.jl:5-33) for BATCHES of
# independent states: MomentumBasedController{N} / addtask! / addcontact! / regularize! /
# setdesired! / disable! keep their names and argument meaning; the per-tick callable takes
# matrices whose columns are the per-state records (Julia column-major == the C ABI's
# "record of QP b is contiguous").
#
# NOTE: no `julia` binary exists in the build image, so this file is exercised only through the
# identical ctypes binding (qpcontrol.jl_amd/binding.py, api.py) -- every ccall below binds one
# symbol of include/qpc.h with the argument order shown there.
module QPControlHIP

export qp_solve_batch, setlaw!, simulate!, SimParams, LAW_INPUT, LAW_CONSTANT, LAW_JOINT_PD, LAW_COM_PD, LAW_ORIENTATION_PD,
       LAW_SE3_PD, SE3_RECORD, load_moi!, pin!, pinned_zeros, enable_warm_start!, control_async!, wait!, desired_range,
       BatchedMomentumBasedController, QPCModel, addtask!, addcontact!, regularize!, finalize!,
       QPSolveFailure, SpatialAccelerationTask, AngularAccelerationTask, LinearAccelerationTask,
       PointAccelerationTask, JointAccelerationTask, MomentumRateTask, LinearMomentumRateTask

const libqpc = get(ENV, "QPCONTROL_HIP_LIB", "libqpcontrol_hip.so")

const QPC_WORLD = Int32(-1)
const QPC_FRAME_DEFAULT = Int32(-2)
@enum TaskKind::Int32 SPATIAL=0 ANGULAR=1 LINEAR=2 POINT=3 JOINT=4 MOMENTUM_RATE=5 LINEAR_MOMENTUM_RATE=6
@enum WeightKind::Int32 HARD=0 SCALAR=1 MATRIX=2 SCALAR_PER_QP=3 MATRIX_PER_QP=4

# exceptions.jl:1-11
# (the reference's three fields -- names of the MOI enum values, from qpc_status_triple -- plus the batch element)
struct QPSolveFailure <: Exception
    terminationstatus::Symbol
    primalstatus::Symbol
    dualstatus::Symbol
    index::Int
    status::Int32
end
function QPSolveFailure(index::Integer, status::Integer)
    t = Ref{Cstring}(C_NULL); p = Ref{Cstring}(C_NULL); d = Ref{Cstring}(C_NULL)
    rc = ccall((:qpc_status_triple, libqpc), Int32, (Int32, Ref{Cstring}, Ref{Cstring}, Ref{Cstring}), Int32(status), t, p, d)
    rc == 0 || return QPSolveFailure(:UNKNOWN, :UNKNOWN_RESULT_STATUS, :UNKNOWN_RESULT_STATUS, Int(index), Int32(status))
    QPSolveFailure(Symbol(unsafe_string(t[])), Symbol(unsafe_string(p[])), Symbol(unsafe_string(d[])), Int(index), Int32(status))
end
Base.showerror(io::IO, e::QPSolveFailure) = print(io, """
    QP solve unsuccessful (batch element $(e.index)).
    Termination status: $(e.terminationstatus)
    Primal status: $(e.primalstatus)
    Dual status: $(e.dualstatus)""")

lasterror() = unsafe_string(ccall((:qpc_last_error, libqpc), Cstring, ()))
check(rc, what) = rc == 0 ? nothing : error("$what failed (code $rc): $(lasterror())")

# ---- qpc_model_desc (include/qpc.h) ----------------------------------------------------------
struct ModelDesc
    nbodies::Int32
    parent::Ptr{Int32}
    joint_type::Ptr{Int32}
    joint_axis::Ptr{Float64}
    joint_R::Ptr{Float64}
    joint_p::Ptr{Float64}
    mass::Ptr{Float64}
    com::Ptr{Float64}
    inertia::Ptr{Float64}
    gravity::NTuple{3, Float64}
end

mutable struct QPCModel
    handle::Ptr{Cvoid}
    nq::Int
    nv::Int
end

"""
    QPCModel(parent, joint_type, axis, R, p, mass, com, inertia, gravity)

Flat tree description == `Mechanism{Float64}` after `remove_fixed_tree_joints!`; with
RigidBodyDynamics loaded, build the arrays by walking `tree_joints(mechanism)`:
`parent[i]` = index of `predecessor(joint_i)`, `R/p` = `joint_to_predecessor(joint_i)`,
`mass/com/inertia` from `spatial_inertia(successor(joint_i))` about its centre of mass.
"""
function QPCModel(parent::Vector{Int32}, joint_type::Vector{Int32}, axis::Matrix{Float64},
                  R::Array{Float64, 3}, p::Matrix{Float64}, mass::Vector{Float64},
                  com::Matrix{Float64}, inertia::Matrix{Float64}, gravity::NTuple{3, Float64})
    nb = length(parent)
    # C side wants row-major 3x3 per body: permute Julia's column-major
    Rrm = permutedims(R, (2, 1, 3))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve parent joint_type axis Rrm p mass com inertia begin
        desc = ModelDesc(nb, pointer(parent), pointer(joint_type), pointer(axis), pointer(Rrm),
                         pointer(p), pointer(mass), pointer(com), pointer(inertia), gravity)
        check(ccall((:qpc_model_create, libqpc), Int32, (Ref{ModelDesc}, Ref{Ptr{Cvoid}}), desc, h),
              "qpc_model_create")
    end
    m = QPCModel(h[], ccall((:qpc_model_nq, libqpc), Int32, (Ptr{Cvoid},), h[]),
                 ccall((:qpc_model_nv, libqpc), Int32, (Ptr{Cvoid},), h[]))
    finalizer(x -> ccall((:qpc_model_destroy, libqpc), Cvoid, (Ptr{Cvoid},), x.handle), m)
    m
end

# ---- tasks: plain descriptors (src/tasks.jl structs without the Parametron machinery) --------
abstract type AbstractMotionTask end
struct SpatialAccelerationTask <: AbstractMotionTask; base::Int32; target::Int32; frame::Int32; end
struct AngularAccelerationTask <: AbstractMotionTask; base::Int32; target::Int32; frame::Int32; end
struct LinearAccelerationTask <: AbstractMotionTask; base::Int32; target::Int32; frame::Int32; end
struct PointAccelerationTask <: AbstractMotionTask; base::Int32; target::Int32; point::NTuple{3, Float64}; end
struct JointAccelerationTask <: AbstractMotionTask; joint::Int32; end
struct MomentumRateTask <: AbstractMotionTask end
struct LinearMomentumRateTask <: AbstractMotionTask end
SpatialAccelerationTask(base, target; frame = QPC_FRAME_DEFAULT) = SpatialAccelerationTask(base, target, frame)
AngularAccelerationTask(base, target; frame = QPC_FRAME_DEFAULT) = AngularAccelerationTask(base, target, frame)
LinearAccelerationTask(base, target; frame = QPC_FRAME_DEFAULT) = LinearAccelerationTask(base, target, frame)

kind(::SpatialAccelerationTask) = SPATIAL; kind(::AngularAccelerationTask) = ANGULAR
kind(::LinearAccelerationTask) = LINEAR; kind(::PointAccelerationTask) = POINT
kind(::JointAccelerationTask) = JOINT; kind(::MomentumRateTask) = MOMENTUM_RATE
kind(::LinearMomentumRateTask) = LINEAR_MOMENTUM_RATE

# ---- controller ------------------------------------------------------------------------------
struct Layout
    nq::Int32; nv::Int32; n_tasks::Int32; n_contacts::Int32; n_basis::Int32; n_desired::Int32
    n_task_weights::Int32; n_surface::Int32; ref_n_var::Int32; ref_n_eq::Int32; ref_n_ineq::Int32
    n_free::Int32; n_pinned::Int32; n_slack::Int32; n_rho::Int32; n_eq_rows::Int32
    lds_bytes_per_qp::Int32; n_desired_inputs::Int32
end

mutable struct BatchedMomentumBasedController{N}
    handle::Ptr{Cvoid}
    model::QPCModel
    layout::Union{Nothing, Layout}
    desired_offsets::Vector{Int}   # task id -> first row in `desired`
    vdot::Matrix{Float64}          # result.v̇ of the last tick (momentum.jl:62-64)
    pointwrenches::Matrix{Float64} # per contact point: 6·nc × B (world frame, about the world origin)
    contactwrenches::Dict{Int32, Matrix{Float64}} # body => 6 × B: the SUM over the body's contact points, what the reference's
                                   # `contactwrenches[BodyID(body)]` holds (momentum.jl:64-72); filled by `contactwrenches!`
    status::Vector{Int32}
end

# MomentumBasedController{N}(mechanism, optimizer; floatingjoint) -- momentum.jl:15-33.
# There is no optimizer argument: the solver lives in the kernels.
function BatchedMomentumBasedController{N}(model::QPCModel; floatingjoint::Integer = QPC_WORLD) where {N}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:qpc_controller_create, libqpc), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{Ptr{Cvoid}}),
                model.handle, N, floatingjoint, h), "qpc_controller_create")
    c = BatchedMomentumBasedController{N}(h[], model, nothing, Int[], zeros(0, 0), zeros(0, 0), Int32[])
    finalizer(x -> ccall((:qpc_controller_destroy, libqpc), Cvoid, (Ptr{Cvoid},), x.handle), c)
    c
end

function _addtask!(c, task, wkind::WeightKind, w)
    base = hasproperty(task, :base) ? task.base : QPC_WORLD
    target = hasproperty(task, :target) ? task.target : QPC_WORLD
    frame = hasproperty(task, :frame) ? task.frame : QPC_FRAME_DEFAULT
    point = hasproperty(task, :point) ? collect(task.point) : zeros(3)
    joint = hasproperty(task, :joint) ? task.joint : QPC_WORLD
    tid = Ref{Int32}(-1)
    wptr = w === nothing ? Ptr{Float64}(C_NULL) : pointer(w)
    GC.@preserve point w check(ccall((:qpc_add_task, libqpc), Int32,
        (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ref{Int32}),
        c.handle, Int32(kind(task)), base, target, frame, point, joint, Int32(wkind), wptr, tid), "qpc_add_task")
    tid[]
end
# the three addtask! methods of momentum.jl:99-117 (+ per-QP "Parameter" weights)
addtask!(c::BatchedMomentumBasedController, task::AbstractMotionTask) = _addtask!(c, task, HARD, nothing)
addtask!(c::BatchedMomentumBasedController, task::AbstractMotionTask, w::Number) = _addtask!(c, task, SCALAR, [Float64(w)])
# W is row-major on the C side: hand over W' so that element (a,b) lands at [a*dim+b]
addtask!(c::BatchedMomentumBasedController, task::AbstractMotionTask, W::AbstractMatrix) =
    _addtask!(c, task, MATRIX, vec(Matrix{Float64}(W')))
addtask!(c::BatchedMomentumBasedController, task::AbstractMotionTask, ::Val{:scalar_per_qp}) = _addtask!(c, task, SCALAR_PER_QP, nothing)
addtask!(c::BatchedMomentumBasedController, task::AbstractMotionTask, ::Val{:matrix_per_qp}) = _addtask!(c, task, MATRIX_PER_QP, nothing)

# regularize! (momentum.jl:128-131)
regularize!(c::BatchedMomentumBasedController, joint::Integer, weight::Real) =
    check(ccall((:qpc_regularize, libqpc), Int32, (Ptr{Cvoid}, Int32, Float64), c.handle, joint, weight), "qpc_regularize")

# regularize!(c, joint, weight::Parameter) (momentum.jl:128-131 takes any `weight`): the weight becomes a per-QP input; returns the
# (0-based) row of `task_weights` that holds it
function regularize!(c::BatchedMomentumBasedController, joint::Integer, ::Val{:per_qp})
    idx = Ref{Int32}(-1)
    check(ccall((:qpc_regularize_per_qp, libqpc), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}), c.handle, joint, idx), "qpc_regularize_per_qp")
    Int(idx[])
end

# arithmetic of the tick kernel (before finalize!): :f64 (default, tau within 1e-8 of the CPU OSQP path) or :f32 (BASELINE config 5's
# variant: the same device source in single precision, tau to ~1e-6 median / 1e-4 at the 99.9th percentile; include/qpc.h)
set_precision!(c::BatchedMomentumBasedController, p::Symbol) =
    check(ccall((:qpc_set_precision, libqpc), Int32, (Ptr{Cvoid}, Int32), c.handle, p === :f32 ? 1 : 0), "qpc_set_precision")
# two QPs per wavefront (opt-in, measured slower on MI355X: DESIGN.md 7c)
set_qps_per_wavefront!(c::BatchedMomentumBasedController, n::Integer) =
    check(ccall((:qpc_set_qps_per_wavefront, libqpc), Int32, (Ptr{Cvoid}, Int32), c.handle, n), "qpc_set_qps_per_wavefront")

# addcontact!(controller, body, position, normal, mu) (momentum.jl:142-148); weight[] and
# maxnormalforce[] (contacts.jl:35-36) are per-QP rows of the tick inputs; disable!(point)
# (contacts.jl:72) == maxnormalforce = 0 for that QP.
# Desired-acceleration laws evaluated on the device (include/qpc.h qpc_task_set_law): what the
# reference's StandingController computes on the CPU before every tick (src/highlevel/standing.jl:58-89).
# `task_id` is the value returned by addtask!.  With a law on every task the tick needs no `desired`.
const LAW_INPUT = Int32(0); const LAW_CONSTANT = Int32(1); const LAW_JOINT_PD = Int32(2)
const LAW_COM_PD = Int32(3); const LAW_ORIENTATION_PD = Int32(4)
# SE3PDController + SE3Trajectory on the device (src/lowlevel/se3pdcontroller.jl:13-18): kp / kd = angular gains,
# ref = [kp_linear, kd_linear]; the task's rows of `desired` become a trajectory record of SE3_RECORD entries
# (include/qpc.h QPC_LAW_SE3_PD; `desired_range(controller, task_id)` tells which rows)
const LAW_SE3_PD = Int32(5); const SE3_RECORD = 29
# ... of a Piecewise trajectory (src/trajectories/piecewise.jl) with np pieces: ref = [kp_linear, kd_linear, np], record of
# se3_pw_record(np) entries = [6 unused | t | np + 1 breaks | np x (x0 xf c0..c5 y0 yf p0 pf)], the piece chosen per QP on the device
const SE3_MAX_PIECES = 4
se3_pw_record(np::Integer) = 8 + 23 * np
function setlaw!(c::BatchedMomentumBasedController, task_id::Integer, law::Int32; kp::Real = 0.0, kd::Real = 0.0,
                 ref::Vector{Float64} = Float64[])
    check(ccall((:qpc_task_set_law, libqpc), Int32, (Ptr{Cvoid}, Int32, Int32, Float64, Float64, Ptr{Float64}, Int32),
                c.handle, task_id, law, kp, kd, ref, length(ref)), "qpc_task_set_law")
    nothing
end

# position / normal / mu may be Parameters in the reference (contacts.jl:39): per_qp_surface (normal, mu) and
# per_qp_position make them rows of the per-QP `surfaces` input ([nx ny nz mu] then [px py pz] per such contact)
function addcontact!(c::BatchedMomentumBasedController, body::Integer, position, normal, mu::Real;
                     per_qp_surface::Bool = false, per_qp_position::Bool = false)
    cid = Ref{Int32}(-1)
    pos, nrm = collect(Float64, position), collect(Float64, normal)
    flags = Int32(per_qp_surface ? 1 : 0) | Int32(per_qp_position ? 2 : 0)
    GC.@preserve pos nrm check(ccall((:qpc_add_contact, libqpc), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Int32, Ref{Int32}),
        c.handle, body, pos, nrm, mu, flags, cid), "qpc_add_contact")
    cid[]
end

# warm-started ticks (the reference's OSQP workspace persists between ticks, momentum.jl:58): call before finalize!;
# the tick then takes / returns `working_set` (2 x B UInt64, zeros = cold start)
enable_warm_start!(c::BatchedMomentumBasedController, on::Bool = true) =
    check(ccall((:qpc_enable_warm_start, libqpc), Int32, (Ptr{Cvoid}, Int32), c.handle, on), "qpc_enable_warm_start")

# initialize! (momentum.jl:150-156): explicit here because the layout is needed to size inputs.
# devices = [0, 1, ...]: the task program is replicated on every device and each batch is split into contiguous
# column ranges, one per device (qpc_finalize_devices) -- one Julia process drives all GPUs of the node.
function finalize!(c::BatchedMomentumBasedController; device::Integer = 0, devices::Union{Nothing, Vector{Int32}} = nothing)
    lay = Ref{Layout}()
    if devices === nothing
        check(ccall((:qpc_finalize, libqpc), Int32, (Ptr{Cvoid}, Int32, Ref{Layout}), c.handle, device, lay), "qpc_finalize")
    else
        check(ccall((:qpc_finalize_devices, libqpc), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ref{Layout}),
                    c.handle, length(devices), devices, lay), "qpc_finalize_devices")
    end
    c.layout = lay[]
    c
end

# rows of `desired` that belong to a task (after finalize!): first row (1-based) and count
function desired_range(c::BatchedMomentumBasedController, task_id::Integer)
    off, len = Ref{Int32}(0), Ref{Int32}(0)
    check(ccall((:qpc_task_desired_range, libqpc), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}, Ref{Int32}), c.handle, task_id, off, len),
          "qpc_task_desired_range")
    (off[] + 1):(off[] + len[])
end

# Page-locked batch arrays: the kernels read / write them IN PLACE over PCIe (no staging copies; measured 89 % of
# the device-resident rate against ~60 % for pageable arrays).  Allocate the batch matrices once, pin them, reuse
# them every tick -- what a control loop does anyway.
function pin!(A::Array)
    check(ccall((:qpc_host_register, libqpc), Int32, (Ptr{Cvoid}, UInt64), pointer(A), sizeof(A)), "qpc_host_register")
    finalizer(x -> ccall((:qpc_host_unregister, libqpc), Int32, (Ptr{Cvoid},), pointer(x)), A)
    A
end
pinned_zeros(::Type{T}, dims...) where {T} = pin!(zeros(T, dims...))

"""
    controller(τ, t, q, v; desired, contact_weight, contact_maxnormalforce, task_weights, surfaces)

The callable of momentum.jl:41-81 for a batch: `q` is nq×B, `v` nv×B, `desired` n_desired×B
(what `setdesired!` would have stored), `τ` nv×B is overwritten and returned.  After the call
`controller.vdot` (== `result.v̇`), `controller.pointwrenches` (6·nc × B, per contact point) and -- after `contactwrenches!(controller)` --
`controller.contactwrenches` (body => 6 × B, the reference's per-body sums) are available.
Throws `QPSolveFailure` for the first element whose status is not OPTIMAL / ALMOST_OPTIMAL
(checkstatus, momentum.jl:83-91) unless `throw_on_failure = false`.
"""
function (c::BatchedMomentumBasedController)(τ::Matrix{Float64}, t::Number, q::Matrix{Float64}, v::Matrix{Float64};
        desired::Matrix{Float64} = zeros(0, size(q, 2)), contact_weight::Matrix{Float64} = zeros(0, size(q, 2)),
        contact_maxnormalforce::Matrix{Float64} = zeros(0, size(q, 2)),
        task_weights::Union{Nothing, Matrix{Float64}} = nothing, surfaces::Union{Nothing, Matrix{Float64}} = nothing,
        throw_on_failure::Bool = true)
    lay = c.layout
    lay === nothing && error("call finalize!(controller) first")
    B = size(q, 2)
    size(c.vdot) == (lay.nv, B) || (c.vdot = zeros(lay.nv, B))
    size(c.pointwrenches) == (6 * lay.n_contacts, B) || (c.pointwrenches = zeros(6 * lay.n_contacts, B))
    length(c.status) == B || (c.status = zeros(Int32, B))
    ptr(x) = x === nothing || isempty(x) ? Ptr{Float64}(C_NULL) : pointer(x)
    GC.@preserve q v desired contact_weight contact_maxnormalforce task_weights surfaces τ check(ccall(
        (:qpc_control_batch, libqpc), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
        c.handle, B, q, v, ptr(desired), ptr(contact_weight), ptr(contact_maxnormalforce), ptr(task_weights),
        ptr(surfaces), τ, c.vdot, ptr(c.pointwrenches), c.status, Ptr{Int32}(C_NULL)), "qpc_control_batch")
    if throw_on_failure
        bad = findfirst(s -> s > 1, c.status)
        bad === nothing || throw(QPSolveFailure(bad, c.status[bad]))
    end
    τ
end

# asynchronous tick: returns once the job is posted (qpc_control_batch_async); every array must stay alive and
# untouched until wait!(controller) -- keep them in pinned, preallocated buffers.  One job in flight per controller.
function control_async!(c::BatchedMomentumBasedController, τ::Matrix{Float64}, q::Matrix{Float64}, v::Matrix{Float64};
        desired::Matrix{Float64} = zeros(0, size(q, 2)), contact_weight::Matrix{Float64} = zeros(0, size(q, 2)),
        contact_maxnormalforce::Matrix{Float64} = zeros(0, size(q, 2)))
    lay = c.layout
    B = size(q, 2)
    size(c.vdot) == (lay.nv, B) || (c.vdot = zeros(lay.nv, B))
    size(c.pointwrenches) == (6 * lay.n_contacts, B) || (c.pointwrenches = zeros(6 * lay.n_contacts, B))
    length(c.status) == B || (c.status = zeros(Int32, B))
    ptr(x) = isempty(x) ? Ptr{Float64}(C_NULL) : pointer(x)
    check(ccall((:qpc_control_batch_async, libqpc), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
        c.handle, B, q, v, ptr(desired), ptr(contact_weight), ptr(contact_maxnormalforce), Ptr{Float64}(C_NULL),
        Ptr{Float64}(C_NULL), τ, c.vdot, ptr(c.pointwrenches), c.status, Ptr{Int32}(C_NULL)), "qpc_control_batch_async")
    nothing
end
wait!(c::BatchedMomentumBasedController) = check(ccall((:qpc_wait, libqpc), Int32, (Ptr{Cvoid},), c.handle), "qpc_wait")
# the error of an async batch nobody waited for (dropped by the next call, which reports its own batch): (code, text), cleared on read
function last_async_error(c::BatchedMomentumBasedController)
    buf = Vector{UInt8}(undef, 512)
    rc = ccall((:qpc_last_async_error, libqpc), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32), c.handle, buf, length(buf))
    (rc, unsafe_string(pointer(buf)))
end

# ---------------------------------------------------------------------------------------------
# f-2: closed loop on the device.  The reference's acceptance run (notebooks/Standing controller.ipynb
# cells 9-11) wraps the controller in `Dynamics(mechanism, controller)`, adds soft contacts and calls
# RigidBodySim `simulate`; `simulate!` below is the batched counterpart: n_steps x (tick -> forward
# dynamics on compliant ground at the controller's contact points -> semi-implicit Euler), columns of
# q / v are robots and are advanced in place.
struct SimParams
    dt::Float64
    ground_z::Float64
    stiffness::Float64
    damping::Float64
    tangential_damping::Float64
end

"""
    simulate!(controller, q, v, n_steps, params; desired, contact_weight, contact_maxnormalforce, ...)

Returns `(τ_last, status_last, bad)`; `bad[b]` counts the steps whose QP failed for robot `b`
(the reference would have thrown `QPSolveFailure` at the first one).
"""
function simulate!(c::BatchedMomentumBasedController, q::Matrix{Float64}, v::Matrix{Float64}, n_steps::Integer,
        params::SimParams; desired::Matrix{Float64} = zeros(0, size(q, 2)),
        contact_weight::Matrix{Float64} = zeros(0, size(q, 2)),
        contact_maxnormalforce::Matrix{Float64} = zeros(0, size(q, 2)),
        task_weights::Union{Nothing, Matrix{Float64}} = nothing, surfaces::Union{Nothing, Matrix{Float64}} = nothing)
    lay = c.layout
    lay === nothing && error("call finalize!(controller) first")
    B = size(q, 2)
    τ = zeros(lay.nv, B); status = zeros(Int32, B); bad = zeros(Int32, B)
    ptr(x) = x === nothing || isempty(x) ? Ptr{Float64}(C_NULL) : pointer(x)
    GC.@preserve q v desired contact_weight contact_maxnormalforce task_weights surfaces check(ccall(
        (:qpc_rollout_batch, libqpc), Int32,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ref{SimParams}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
        c.handle, B, n_steps, q, v, ptr(desired), ptr(contact_weight), ptr(contact_maxnormalforce), ptr(task_weights),
        ptr(surfaces), Ref(params), τ, status, bad), "qpc_rollout_batch")
    τ, status, bad
end

# ---------------------------------------------------------------------------------------------
# b-1: solver-level plug.  `MomentumBasedController{N}(mechanism, optimizer)` (reference
# src/lowlevel/momentum.jl:15-16,27) accepts any MOI.AbstractOptimizer; the reference's tests build
# an OSQP one (test/runtests.jl:35-43).  `qp_solve_batch` is the ccall that `QPCOptimizer`
# (QPControlHIPMOI.jl, loaded by `load_moi!()`) forwards `MOI.optimize!` to with B = 1: dense OSQP-form data in, (x, y, status) out.
struct Settings
    max_active_set_iters::Int32
    max_refine_iters::Int32
    eps_primal_reg::Float64
    eps_dual_reg::Float64
    tol_residual::Float64
    tol_bound::Float64
end

function default_settings()
    s = Ref(Settings(0, 0, 0.0, 0.0, 0.0, 0.0))
    ccall((:qpc_default_settings, libqpc), Int32, (Ptr{Settings},), s)
    s[]
end

"""
    qp_solve_batch(P, q, A, l, u; device = 0, settings = default_settings())

`min ½x'Px + q'x  s.t. l ≤ Ax ≤ u` for B problems.  Array layouts: `P[n, n, B]` (symmetric),
`q[n, B]`, `A[n, m, B]` (i.e. Aᵀ per problem: the ABI is row-major per QP), `l[m, B]`, `u[m, B]`.
Returns `(x[n, B], y[m, B], status[B], iters[B])`, multipliers in OSQP's sign convention.
"""
function qp_solve_batch(P::Array{Float64, 3}, q::Matrix{Float64}, At::Array{Float64, 3}, l::Matrix{Float64},
                        u::Matrix{Float64}; device::Integer = 0, settings::Settings = default_settings())
    n, B = size(q); m = size(l, 1)
    x = zeros(n, B); y = zeros(max(m, 1), B); status = zeros(Int32, B); iters = zeros(Int32, B)
    rc = ccall((:qpc_qp_solve_batch, libqpc), Int32,
               (Int32, Int32, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                Ref{Settings}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
               n, m, B, P, q, At, l, u, Ref(settings), device, x, y, status, iters)
    rc == 0 || error(unsafe_string(ccall((:qpc_last_error, libqpc), Cstring, ())))
    x, y[1:m, :], status, iters
end

"""
    QpProblem(n, m, Prows, Pcols, Arows, Acols; device = 0)

L0 with the optimizer's life cycle (`qpc_qp_setup` / `qpc_qp_solve` / `qpc_qp_free`, include/qpc.h): the sparsity pattern once
(1-based COO here, 0-based at the ABI; of `P` one triangle), then `solve!(prob, Pvals, q, Avals, l, u; warm_start)` every tick with
the values in the pattern's order -- 13 KB instead of 368 KB for the 143 x 178 Atlas problem, the working set warm-started from the
previous call's multipliers (what OSQP's warm start does for the reference, momentum.jl:58).
"""
mutable struct QpProblem
    handle::Ptr{Cvoid}
    n::Int
    m::Int
    nnzP::Int
    nnzA::Int
    function QpProblem(n::Integer, m::Integer, Prows::Vector{<:Integer}, Pcols::Vector{<:Integer}, Arows::Vector{<:Integer},
                       Acols::Vector{<:Integer}; device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        pr = Int32.(Prows .- 1); pc = Int32.(Pcols .- 1); ar = Int32.(Arows .- 1); ac = Int32.(Acols .- 1)
        rc = ccall((:qpc_qp_setup, libqpc), Int32,
                   (Int32, Int32, Int64, Ptr{Int32}, Ptr{Int32}, Int64, Ptr{Int32}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
                   n, m, length(pr), pr, pc, length(ar), ar, ac, device, h)
        rc == 0 || error(unsafe_string(ccall((:qpc_last_error, libqpc), Cstring, ())))
        p = new(h[], n, m, length(pr), length(ar))
        finalizer(q -> (q.handle == C_NULL || ccall((:qpc_qp_free, libqpc), Cvoid, (Ptr{Cvoid},), q.handle); q.handle = C_NULL), p)
        p
    end
end
"""
    analyze!(prob, l, u) -> plan_info(prob)

The symbolic analysis of the sparse workgroup solver (`qpc_qp_analyze`) for the equality rows of one QP's bounds `l[m]`, `u[m]`: an
optimizer calls it once after `MOI.copy_to` -- the constraint sets are known then -- so that the first tick is not the slow one.
"""
function analyze!(p::QpProblem, l::Vector{Float64}, u::Vector{Float64})
    rc = ccall((:qpc_qp_analyze, libqpc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.handle, l, u)
    rc == 0 || error(unsafe_string(ccall((:qpc_last_error, libqpc), Cstring, ())))
    plan_info(p)
end
"(plan in use, equality rows, other rows, fronts, levels, update jobs, store doubles, LDS bytes, widest front, tallest front, multiply-adds, dense multiply-adds)"
function plan_info(p::QpProblem)
    out = zeros(Int64, 12)
    rc = ccall((:qpc_qp_plan_info, libqpc), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int32), p.handle, out, length(out))
    rc == 0 || error(unsafe_string(ccall((:qpc_last_error, libqpc), Cstring, ())))
    out
end
"`Pvals[nnzP, B]`, `q[n, B]`, `Avals[nnzA, B]`, `l[m, B]`, `u[m, B]` -> `(x[n, B], y[m, B], status[B], iters[B])`"
function solve!(p::QpProblem, Pvals::Matrix{Float64}, q::Matrix{Float64}, Avals::Matrix{Float64}, l::Matrix{Float64},
                u::Matrix{Float64}; warm_start::Bool = false, settings::Settings = default_settings())
    B = size(q, 2)
    x = zeros(p.n, B); y = zeros(max(p.m, 1), B); status = zeros(Int32, B); iters = zeros(Int32, B)
    rc = ccall((:qpc_qp_solve, libqpc), Int32,
               (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Settings}, Int32,
                Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
               p.handle, B, Pvals, q, Avals, l, u, Ref(settings), warm_start ? 1 : 0, x, y, status, iters)
    rc == 0 || error(unsafe_string(ccall((:qpc_last_error, libqpc), Cstring, ())))
    x, y[1:p.m, :], status, iters
end

"""
    solve_device!(p, B, Pvals, q, Avals, l, u, x, y, status, iters; warm_start = false, stream = C_NULL)

`qpc_qp_solve_device`: the same solve on DEVICE pointers (e.g. `pointer(::ROCArray)` of AMDGPU.jl) -- values, bounds and results stay in HBM;
work is queued on `stream` (a `hipStream_t`), the call returns once it has synchronised it.  The plan must exist (`analyze!`).
"""
function solve_device!(p::QpProblem, B::Integer, Pvals::Ptr{Float64}, q::Ptr{Float64}, Avals::Ptr{Float64}, l::Ptr{Float64}, u::Ptr{Float64},
                       x::Ptr{Float64}, y::Ptr{Float64}, status::Ptr{Int32}, iters::Ptr{Int32}; warm_start::Bool = false,
                       stream::Ptr{Cvoid} = C_NULL, settings::Settings = default_settings())
    rc = ccall((:qpc_qp_solve_device, libqpc), Int32,
               (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Settings}, Int32,
                Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Cvoid}),
               p.handle, B, Pvals, q, Avals, l, u, Ref(settings), warm_start ? 1 : 0, x, y, status, iters, stream)
    rc == 0 || error(unsafe_string(ccall((:qpc_last_error, libqpc), Cstring, ())))
    nothing
end

# ---------------------------------------------------------------------------------------------
# f-3: `QPCOptimizer <: MOI.AbstractOptimizer` lives in QPControlHIPMOI.jl (sub-module QPControlHIPMOI), loaded on
# demand: it is the only part of the shim that needs MathOptInterface, it was written against the MOI 0.8 API
# without a Julia toolchain, and a mismatch there must not take `using QPControlHIP` down with it.
"""
    load_moi!() -> Module

Include the experimental `QPCOptimizer <: MOI.AbstractOptimizer` (QPControlHIPMOI.jl next to this file) into this module
and return the sub-module; `QPControlHIP.load_moi!().QPCOptimizer()` is what `MomentumBasedController{N}(mechanism,
optimizer)` (reference momentum.jl:15-16) takes.  Requires MathOptInterface in the active environment.
"""
function load_moi!()
    isdefined(@__MODULE__, :QPControlHIPMOI) || Base.include(@__MODULE__, joinpath(@__DIR__, "QPControlHIPMOI.jl"))
    getfield(@__MODULE__, :QPControlHIPMOI)
end

end # module
