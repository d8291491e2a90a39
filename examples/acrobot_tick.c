/* acrobot_tick.c -- the C ABI of include/qpc.h from plain C99, no Python, no Julia: BASELINE config 1
 * ("Acrobot (2-DoF), single JointAccelerationTask, 1 QP").
 *
 * Builds the Acrobot of the reference's PointAccelerationTask demo (notebooks/PointAccelerationTask Demo.ipynb
 * cells 3-9) through qpc_model_create, registers the controller the way
 * MomentumBasedController(mechanism, optimizer) / addtask! / regularize! do (momentum.jl:15-33, 99-131), runs one tick
 * and checks it against the CLOSED FORM of this 2-link pendulum written out below -- the textbook manipulator
 * equation, independent of every line of the library:
 *     QP:   min 1e-6 |vdot|^2  s.t.  vdot_elbow = d          =>  vdot = (0, d)
 *     tau = M(q) vdot + c(q, qdot) + g(q)                        (inverse_dynamics!, momentum.jl:75)
 *
 *   cc -std=c99 -I include examples/acrobot_tick.c -o acrobot_tick -L qpcontrol.jl_amd -lqpcontrol_hip -lm
 * exit code 0 and the line "ACROBOT_OK" on success. */
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

int main(void) {
  /* two bodies: upper link on the world through `shoulder`, lower link on it through `elbow`, both revolute about y */
  const int32_t parent[2] = {QPC_WORLD, 0};
  const int32_t jtype[2] = {QPC_JOINT_REVOLUTE, QPC_JOINT_REVOLUTE};
  const double axis[6] = {0, 1, 0, 0, 1, 0};
  const double R[18] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double p[6] = {0, 0.15, 0, 0, 0.1, -1.0};
  const double mass[2] = {1.0, 1.0};
  const double com[6] = {0, 0, -0.5, 0, 0, -1.0};
  const double inertia[12] = {0.083, 0.083, 0.001, 0, 0, 0, 0.33, 0.33, 0.001, 0, 0, 0};
  qpc_model_desc desc;
  desc.nbodies = 2; desc.parent = parent; desc.joint_type = jtype; desc.joint_axis = axis; desc.joint_R = R; desc.joint_p = p;
  desc.mass = mass; desc.com = com; desc.inertia = inertia;
  desc.gravity[0] = 0.0; desc.gravity[1] = 0.0; desc.gravity[2] = -9.81;

  qpc_model *model = NULL;
  qpc_controller *ctrl = NULL;
  qpc_layout lay;
  int32_t task = -1;
  CHECK(qpc_model_create(&desc, &model));
  CHECK(qpc_controller_create(model, 4, QPC_WORLD, &ctrl));           /* floatingjoint = nothing */
  CHECK(qpc_add_task(ctrl, QPC_TASK_JOINT_ACCELERATION, QPC_WORLD, QPC_WORLD, QPC_FRAME_DEFAULT, NULL, 1,
                     QPC_WEIGHT_HARD, NULL, &task));                  /* addtask!(controller, JointAccelerationTask(elbow)) */
  CHECK(qpc_regularize(ctrl, 0, 1e-6));                               /* regularize!(controller, joint, 1e-6) for both joints */
  CHECK(qpc_regularize(ctrl, 1, 1e-6));
  CHECK(qpc_finalize(ctrl, 0, &lay));
  if (lay.nq != 2 || lay.nv != 2 || lay.n_desired != 1) { fprintf(stderr, "unexpected layout\n"); return 3; }

  /* one QP: the state of the notebook (cell 9) with some velocity, desired elbow acceleration d */
  const double q[2] = {-3.14159265358979323846, 0.1}, v[2] = {0.3, -0.2}, d = 0.7;
  double tau[2] = {0, 0}, vdot[2] = {0, 0};
  int32_t status = -1, iters = -1;
  CHECK(qpc_control_batch(ctrl, 1, q, v, &d, NULL, NULL, NULL, NULL, tau, vdot, NULL, &status, &iters));

  /* closed form: planar double pendulum about y, angles from the downward vertical */
  const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 1.0, I1 = 0.083, I2 = 0.33, g = 9.81;
  const double s1 = sin(q[0]), s2 = sin(q[1]), c2 = cos(q[1]), s12 = sin(q[0] + q[1]);
  const double M11 = I1 + I2 + m1 * lc1 * lc1 + m2 * (l1 * l1 + lc2 * lc2 + 2.0 * l1 * lc2 * c2);
  const double M12 = I2 + m2 * (lc2 * lc2 + l1 * lc2 * c2), M22 = I2 + m2 * lc2 * lc2;
  const double h = m2 * l1 * lc2 * s2;
  const double vd_ref[2] = {0.0, d};
  const double tau_ref[2] = {
      M11 * vd_ref[0] + M12 * vd_ref[1] - h * (2.0 * v[0] * v[1] + v[1] * v[1]) + m1 * g * lc1 * s1 + m2 * g * (l1 * s1 + lc2 * s12),
      M12 * vd_ref[0] + M22 * vd_ref[1] + h * v[0] * v[0] + m2 * g * lc2 * s12};
  double err = 0.0;
  for (int i = 0; i < 2; i++) {
    err = fmax(err, fabs(tau[i] - tau_ref[i]));
    err = fmax(err, fabs(vdot[i] - vd_ref[i]));
  }
  printf("status %s, vdot = (%.3e, %.12f), tau = (%.12f, %.12f), closed form (%.12f, %.12f), max error %.3e\n",
         qpc_status_string(status), vdot[0], vdot[1], tau[0], tau[1], tau_ref[0], tau_ref[1], err);
  qpc_controller_destroy(ctrl);
  qpc_model_destroy(model);
  if (status != QPC_STATUS_OPTIMAL || !(err < 1e-9)) return 1;
  printf("ACROBOT_OK\n");
  return 0;
}
