# ---------------------------------------------------------------------------------------------
# f-3: `QPCOptimizer <: MOI.AbstractOptimizer` -- lets UNMODIFIED QPControl.jl run its QP on the GPU:
#
#     QPControlHIP.load_moi!()      # includes this file; needs MathOptInterface in the environment
#     controller = MomentumBasedController{4}(mechanism, QPControlHIP.QPControlHIPMOI.QPCOptimizer(), floatingjoint = ...)
#
# Parametron (0.9, MathOptInterface 0.8) drives an optimizer like this (momentum.jl:27,58):
#   * once:        MOI.copy_to(optimizer, backend_model)         -- variables, ONE ScalarQuadraticFunction objective,
#                  VectorAffineFunction-in-{Zeros, Nonnegatives, Nonpositives} constraints
#   * every tick:  MOI.set(optimizer, MOI.ObjectiveFunction{F}(), f)   and
#                  MOI.set(optimizer, MOI.ConstraintFunction(), ci, f)  -- same terms, new coefficients
#                  MOI.optimize!(optimizer)
#   * afterwards:  MOI.get(optimizer, MOI.VariablePrimal(), vi), MOI.TerminationStatus(), MOI.PrimalStatus(),
#                  MOI.DualStatus() (checkstatus, momentum.jl:83-91)
# The optimizer keeps the coefficients dense on the host (P n x n, q, A' n x m, l, u: the MOI terms are added into them) and the
# PATTERN of the terms as a prepared problem of the C ABI (QpProblem: qpc_qp_setup once after copy_to, again only if a term
# appears outside the pattern); optimize! gathers the values in the pattern's order and calls qpc_qp_solve with B = 1,
# warm-started from the previous tick (OSQP's warm start, which the reference relies on: momentum.jl:58 re-solves the same
# model).  Assembly stays on the CPU here: this is the compatibility plug (SURVEY.md 8 b-1); the performance path is
# BatchedMomentumBasedController.
#
# Written against the MOI 0.8 API from memory -- there is no `julia` in the build image, so this block is
# UNTESTED there; its Python twin (qpcontrol.jl_amd/moi.py) replays the same life cycle against the same
# L0 entry in tests/test_moi_optimizer.py.  MathOptInterface (already a dependency of QPControl.jl, Project.toml)
# is the only package this file needs.  EXPERIMENTAL until it has run under Julia: it lives in its own file and
# sub-module so that `using QPControlHIP` (the batched shim) neither needs MathOptInterface nor breaks on an API mismatch here.
module QPControlHIPMOI

export QPCOptimizer

import MathOptInterface
const MOI = MathOptInterface
using LinearAlgebra: dot
import ..QPControlHIP: qp_solve_batch, QpProblem, solve!, Settings, default_settings

mutable struct QPCOptimizer <: MOI.AbstractOptimizer
    device::Int32
    settings::Settings
    n::Int
    m::Int
    P::Matrix{Float64}          # n x n symmetric
    q::Vector{Float64}
    At::Matrix{Float64}         # n x m (column j = row j of A): row-major A for the C ABI
    l::Vector{Float64}
    u::Vector{Float64}
    objconstant::Float64
    sense::MOI.OptimizationSense
    rowranges::Dict{Any, UnitRange{Int}}   # constraint index -> rows of A
    rowsets::Dict{Any, Any}                 # constraint index -> its set (Zeros / Nonnegatives / Nonpositives)
    x::Vector{Float64}
    y::Vector{Float64}
    status::Int32                           # QPC_STATUS_* of the last optimize!, -1 before
    patP::BitMatrix                         # entries of P / A' that a term has ever touched (the sparsity pattern)
    patAt::BitMatrix
    problem::Union{Nothing, QpProblem}      # the prepared problem of that pattern (nothing: set up at the next optimize!)
    iP::Vector{Int}; jP::Vector{Int}; iA::Vector{Int}; jA::Vector{Int}   # its COO (P: upper triangle; A: row, column)
    solved::Int                             # optimize! calls on this problem (the first one starts cold)
    QPCOptimizer(; device::Integer = 0, settings::Settings = default_settings()) =
        new(device, settings, 0, 0, zeros(0, 0), Float64[], zeros(0, 0), Float64[], Float64[], 0.0,
            MOI.MIN_SENSE, Dict{Any, UnitRange{Int}}(), Dict{Any, Any}(), Float64[], Float64[], Int32(-1),
            falses(0, 0), falses(0, 0), nothing, Int[], Int[], Int[], Int[], 0)
end

MOI.get(::QPCOptimizer, ::MOI.SolverName) = "qpcontrol_hip (gfx950)"
MOI.is_empty(o::QPCOptimizer) = o.n == 0
function MOI.empty!(o::QPCOptimizer)
    o.n = o.m = 0; o.P = zeros(0, 0); o.q = Float64[]; o.At = zeros(0, 0); o.l = Float64[]; o.u = Float64[]
    o.objconstant = 0.0; empty!(o.rowranges); empty!(o.rowsets); o.x = Float64[]; o.y = Float64[]; o.status = Int32(-1)
    o.patP = falses(0, 0); o.patAt = falses(0, 0); o.problem = nothing; o.solved = 0
    nothing
end

const _QPCSets = Union{MOI.Zeros, MOI.Nonnegatives, MOI.Nonpositives}
MOI.supports(::QPCOptimizer, ::MOI.ObjectiveSense) = true
MOI.supports(::QPCOptimizer, ::MOI.ObjectiveFunction{MOI.ScalarQuadraticFunction{Float64}}) = true
MOI.supports(::QPCOptimizer, ::MOI.ObjectiveFunction{MOI.ScalarAffineFunction{Float64}}) = true
MOI.supports_constraint(::QPCOptimizer, ::Type{MOI.VectorAffineFunction{Float64}}, ::Type{<:_QPCSets}) = true

# rows of one vector constraint: A x + b in set  ->  l <= A x <= u
function _setrows!(o::QPCOptimizer, rows::UnitRange{Int}, f::MOI.VectorAffineFunction{Float64}, set)
    o.At[:, rows] .= 0.0
    for t in f.terms
        i, r = t.scalar_term.variable_index.value, rows[t.output_index]
        o.At[i, r] += t.scalar_term.coefficient
        o.patAt[i, r] || (o.patAt[i, r] = true; o.problem = nothing)   # a term outside the pattern: set up again
    end
    for (k, r) in enumerate(rows)
        b = f.constants[k]
        if set isa MOI.Zeros
            o.l[r] = -b; o.u[r] = -b
        elseif set isa MOI.Nonnegatives
            o.l[r] = -b; o.u[r] = Inf
        else
            o.l[r] = -Inf; o.u[r] = -b
        end
    end
end
function _setobjective!(o::QPCOptimizer, f::MOI.ScalarQuadraticFunction{Float64})
    o.P .= 0.0; o.q .= 0.0
    for t in f.quadratic_terms   # MOI: 1/2 x'Qx; an off-diagonal term appears once and stands for both entries
        i, j = t.variable_index_1.value, t.variable_index_2.value
        o.P[i, j] += t.coefficient
        i == j || (o.P[j, i] += t.coefficient)
        a, b = minmax(i, j)
        o.patP[a, b] || (o.patP[a, b] = true; o.problem = nothing)
    end
    for t in f.affine_terms
        o.q[t.variable_index.value] += t.coefficient
    end
    o.objconstant = f.constant
end
function _setobjective!(o::QPCOptimizer, f::MOI.ScalarAffineFunction{Float64})
    o.P .= 0.0; o.q .= 0.0
    for t in f.terms
        o.q[t.variable_index.value] += t.coefficient
    end
    o.objconstant = f.constant
end

# MOI.copy_to: variables keep their indices (Parametron creates them 1:n), constraints are laid out in
# the order of ListOfConstraints / ListOfConstraintIndices
function MOI.copy_to(dest::QPCOptimizer, src::MOI.ModelLike; copy_names::Bool = false)
    MOI.empty!(dest)
    vis = MOI.get(src, MOI.ListOfVariableIndices())
    n = length(vis)
    idxmap = MOI.Utilities.IndexMap()
    for (k, vi) in enumerate(vis)
        idxmap[vi] = MOI.VariableIndex(k)
    end
    cons = Any[]
    m = 0
    for (F, S) in MOI.get(src, MOI.ListOfConstraints())
        (F == MOI.VectorAffineFunction{Float64} && S <: _QPCSets) || throw(MOI.UnsupportedConstraint{F, S}())
        for ci in MOI.get(src, MOI.ListOfConstraintIndices{F, S}())
            f = MOI.get(src, MOI.ConstraintFunction(), ci)
            s = MOI.get(src, MOI.ConstraintSet(), ci)
            push!(cons, (ci, f, s, (m + 1):(m + MOI.dimension(s))))
            m += MOI.dimension(s)
        end
    end
    dest.n = n; dest.m = m
    dest.P = zeros(n, n); dest.q = zeros(n); dest.At = zeros(n, m); dest.l = zeros(m); dest.u = zeros(m)
    dest.patP = falses(n, n); dest.patAt = falses(n, m)
    for (ci, f, s, rows) in cons
        fm = MOI.Utilities.mapvariables(idxmap, f)
        dest.rowranges[ci] = rows; dest.rowsets[ci] = s
        idxmap[ci] = ci
        _setrows!(dest, rows, fm, s)
    end
    dest.sense = MOI.get(src, MOI.ObjectiveSense())
    dest.sense == MOI.MAX_SENSE && error("QPCOptimizer: minimisation only (momentum.jl:158-160 uses Minimize)")
    F = MOI.get(src, MOI.ObjectiveFunctionType())
    _setobjective!(dest, MOI.Utilities.mapvariables(idxmap, MOI.get(src, MOI.ObjectiveFunction{F}())))
    idxmap
end

# per-tick coefficient updates (Parametron update!)
MOI.set(o::QPCOptimizer, ::MOI.ObjectiveFunction{F}, f::F) where {F <: Union{MOI.ScalarQuadraticFunction{Float64}, MOI.ScalarAffineFunction{Float64}}} =
    _setobjective!(o, f)
MOI.set(o::QPCOptimizer, ::MOI.ObjectiveSense, s::MOI.OptimizationSense) = (o.sense = s)
function MOI.set(o::QPCOptimizer, ::MOI.ConstraintFunction, ci::MOI.ConstraintIndex, f::MOI.VectorAffineFunction{Float64})
    _setrows!(o, o.rowranges[ci], f, o.rowsets[ci])
end
MOI.set(o::QPCOptimizer, ::MOI.ConstraintSet, ci::MOI.ConstraintIndex, s) = (o.rowsets[ci] = s)

function MOI.optimize!(o::QPCOptimizer)
    lc = clamp.(o.l, -1e20, 1e20); uc = clamp.(o.u, -1e20, 1e20)   # +-1e20 = unbounded (include/qpc.h L0)
    if o.m == 0   # no constraints: the dense entry
        x, y, status, _ = qp_solve_batch(reshape(o.P, o.n, o.n, 1), reshape(o.q, o.n, 1), reshape(o.At, o.n, o.m, 1),
                                         reshape(lc, o.m, 1), reshape(uc, o.m, 1); device = o.device, settings = o.settings)
        o.x = vec(x); o.y = vec(y); o.status = status[1]
        return nothing
    end
    if o.problem === nothing   # MOI.copy_to has just run (or the pattern grew): qpc_qp_setup
        cP = findall(o.patP); cA = findall(o.patAt)
        o.iP = [c[1] for c in cP]; o.jP = [c[2] for c in cP]
        o.jA = [c[1] for c in cA]; o.iA = [c[2] for c in cA]          # patAt is A' (variable, row)
        o.problem = QpProblem(o.n, o.m, o.iP, o.jP, o.iA, o.jA; device = o.device)
        analyze!(o.problem, lc, uc)   # the plan of the sparse workgroup solver for these constraint sets (rows with l == u: MOI.Zeros)
        o.solved = 0
    end
    Pv = Float64[o.P[i, j] for (i, j) in zip(o.iP, o.jP)]
    Av = Float64[o.At[j, i] for (i, j) in zip(o.iA, o.jA)]
    x, y, status, _ = solve!(o.problem, reshape(Pv, :, 1), reshape(o.q, o.n, 1), reshape(Av, :, 1), reshape(lc, o.m, 1),
                             reshape(uc, o.m, 1); warm_start = o.solved > 0, settings = o.settings)
    o.solved += 1
    o.x = vec(x); o.y = vec(y); o.status = status[1]
    nothing
end

# results: exactly what checkstatus reads (momentum.jl:83-91)
function MOI.get(o::QPCOptimizer, ::MOI.TerminationStatus)
    s = o.status
    s == -1 ? MOI.OPTIMIZE_NOT_CALLED : s == 0 ? MOI.OPTIMAL : s == 1 ? MOI.ALMOST_OPTIMAL : s == 2 ? MOI.INFEASIBLE :
    s == 3 ? MOI.DUAL_INFEASIBLE : s == 4 ? MOI.ITERATION_LIMIT : MOI.NUMERICAL_ERROR
end
MOI.get(o::QPCOptimizer, ::MOI.PrimalStatus) =
    o.status == 0 ? MOI.FEASIBLE_POINT : o.status == 2 ? MOI.NO_SOLUTION : o.status == 3 ? MOI.INFEASIBILITY_CERTIFICATE :
    o.status == -1 ? MOI.NO_SOLUTION : MOI.UNKNOWN_RESULT_STATUS
MOI.get(o::QPCOptimizer, ::MOI.DualStatus) =
    o.status == 0 ? MOI.FEASIBLE_POINT : o.status == 2 ? MOI.INFEASIBILITY_CERTIFICATE : o.status == -1 ? MOI.NO_SOLUTION :
    MOI.UNKNOWN_RESULT_STATUS
MOI.get(o::QPCOptimizer, ::MOI.ResultCount) = o.status == -1 ? 0 : 1
MOI.get(o::QPCOptimizer, ::MOI.VariablePrimal, vi::MOI.VariableIndex) = o.x[vi.value]
MOI.get(o::QPCOptimizer, ::MOI.ObjectiveValue) = 0.5 * dot(o.x, o.P * o.x) + dot(o.q, o.x) + o.objconstant
MOI.get(o::QPCOptimizer, ::MOI.ConstraintDual, ci::MOI.ConstraintIndex) = -o.y[o.rowranges[ci]]  # OSQP sign convention
MOI.get(o::QPCOptimizer, ::MOI.ConstraintPrimal, ci::MOI.ConstraintIndex) = (o.At' * o.x)[o.rowranges[ci]]
end # module QPControlHIPMOI
