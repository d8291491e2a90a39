/* qp_optimizer_cycle.c -- the solver-level plug of include/qpc.h from plain C99: what an `MOI.AbstractOptimizer` does for the
 * reference (Parametron.Model(optimizer), /root/reference/src/lowlevel/momentum.jl:15-16,27,58; built as in
 * test/runtests.jl:35-43), with the optimizer's life cycle at the ABI:
 *     qpc_qp_setup   <- MOI.copy_to        (the sparsity pattern, once)
 *     qpc_qp_solve   <- MOI.optimize!      (new coefficients every tick, warm-started from the previous one)
 *     qpc_qp_free    <- MOI.empty!
 * The QP is small enough to check by hand -- a point pulled towards a moving target inside a box, tied to a plane:
 *     min  1/2 (2 x0^2 + 2 x1^2 + 2 x2^2) - 2 t'x      s.t.   x0 + x1 + x2 = 1 ,   0 <= x_i <= 0.6
 * i.e. the projection of t onto {sum = 1} intersected with the box.  Five "ticks" move t; the closed form below (clip a shifted t,
 * the shift found by bisection on the plane equation) is independent of the library.
 *
 *   cc -std=c99 -I include examples/qp_optimizer_cycle.c -o qp_cycle -L qpcontrol.jl_amd -lqpcontrol_hip -lm
 * exit code 0 and the line "QP_CYCLE_OK" on success. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qpc.h"

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int32_t rc__ = (call);                                                                   \
    if (rc__ != QPC_OK) {                                                                    \
      fprintf(stderr, "%s failed (%d): %s\n", #call, (int)rc__, qpc_last_error());          \
      return 2;                                                                              \
    }                                                                                        \
  } while (0)

static double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* projection of t onto {sum x = 1, 0 <= x <= 0.6}: x_i = clip(t_i - nu, 0, 0.6) with nu from the plane equation */
static void closed_form(const double t[3], double x[3]) {
  double lo = -10.0, hi = 10.0;
  int it, i;
  for (it = 0; it < 200; it++) {
    const double nu = 0.5 * (lo + hi);
    double s = 0.0;
    for (i = 0; i < 3; i++) s += clip(t[i] - nu, 0.0, 0.6);
    if (s > 1.0) lo = nu; else hi = nu;
  }
  for (i = 0; i < 3; i++) x[i] = clip(t[i] - 0.5 * (lo + hi), 0.0, 0.6);
}

int main(void) {
  /* pattern: P diagonal (3 entries); A = [1 1 1; I] (6 entries), 0-based COO */
  const int32_t P_rows[3] = {0, 1, 2}, P_cols[3] = {0, 1, 2};
  const int32_t A_rows[6] = {0, 0, 0, 1, 2, 3}, A_cols[6] = {0, 1, 2, 0, 1, 2};
  const double P_vals[3] = {2.0, 2.0, 2.0}, A_vals[6] = {1, 1, 1, 1, 1, 1};
  const double l[4] = {1.0, 0.0, 0.0, 0.0}, u[4] = {1.0, 0.6, 0.6, 0.6};
  qpc_qp_problem *prob = NULL;
  double worst = 0.0;
  int tick, i;
  CHECK(qpc_qp_setup(3, 4, 3, P_rows, P_cols, 6, A_rows, A_cols, 0, &prob));
  for (tick = 0; tick < 5; tick++) {
    const double t[3] = {0.9 - 0.2 * tick, 0.1 + 0.15 * tick, 0.2 + 0.05 * tick};
    double q[3], x[3], y[4], ref[3];
    int32_t status = -1, iters = -1;
    for (i = 0; i < 3; i++) q[i] = -2.0 * t[i];
    CHECK(qpc_qp_solve(prob, 1, P_vals, q, A_vals, l, u, NULL, tick > 0, x, y, &status, &iters));
    if (status != QPC_STATUS_OPTIMAL) { fprintf(stderr, "tick %d: status %s\n", tick, qpc_status_string(status)); return 1; }
    closed_form(t, ref);
    for (i = 0; i < 3; i++) worst = fmax(worst, fabs(x[i] - ref[i]));
    printf("tick %d: x = (%.6f %.6f %.6f)  closed form (%.6f %.6f %.6f)  working-set changes %d\n", tick, x[0], x[1], x[2], ref[0], ref[1],
           ref[2], (int)iters);
  }
  qpc_qp_free(prob);
  if (worst > 1e-8) { fprintf(stderr, "max |x - closed form| = %.3e\n", worst); return 1; }
  printf("QP_CYCLE_OK max |x - closed form| = %.2e\n", worst);
  return 0;
}
