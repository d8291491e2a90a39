/* ora_standing.c -- CPU restatement of the PD laws of the reference's StandingController tick,
 * (controller::StandingController)(tau, t, state), src/highlevel/standing.jl:58-89: the caller of
 * the hot path that produces the `desired` record before every MomentumBasedController tick.
 * TEST INFRASTRUCTURE ONLY (see oracle.h): never linked into or imported by the product.
 *
 * RigidBodyDynamics.PDControl semantics (un-vendored dependency, RBD 2.x):
 *   pd(gains, e, edot)            = -k e - d edot
 *   pd(gains, x, xdes, v, vdes)   = pd(gains, x - xdes, v - vdes)
 *   pd(gains, R::Rotation, omega) = pd(gains, rotation_vector(R), omega)   (error = log map of R)
 */
#include <math.h>
#include <string.h>

#include "oracle.h"

/* standing.jl:60-67: e = c - c_ref, edot = linear(momentum(state)) / m, setdesired!(linmomtask, m * pd) */
void ora_standing_com_law(const ora_model *m, const ora_state *s, const double *v, double kp, double kd,
                          const double comref[3], double out[3]) {
  for (int a = 0; a < 3; a++) {
    double hl = 0.0; /* linear momentum: world-origin / centroidal frames differ by a translation only */
    for (int k = 0; k < m->nv; k++) hl += s->Amom[3 + a][k] * v[k];
    double e = s->com[a] - comref[a], edot = hl / m->mtot;
    out[a] = m->mtot * (-kp * e - kd * edot);
  }
}

/* log map via the unit quaternion of R (Shepperd's method), independent of the kernel's formula */
static void rotvec_from_R(const double R[9], double rv[3]) {
  double q[4]; /* w x y z */
  double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    double S = sqrt(tr + 1.0) * 2.0;
    q[0] = 0.25 * S; q[1] = (R[7] - R[5]) / S; q[2] = (R[2] - R[6]) / S; q[3] = (R[3] - R[1]) / S;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    double S = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
    q[0] = (R[7] - R[5]) / S; q[1] = 0.25 * S; q[2] = (R[1] + R[3]) / S; q[3] = (R[2] + R[6]) / S;
  } else if (R[4] > R[8]) {
    double S = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
    q[0] = (R[2] - R[6]) / S; q[1] = (R[1] + R[3]) / S; q[2] = 0.25 * S; q[3] = (R[5] + R[7]) / S;
  } else {
    double S = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
    q[0] = (R[3] - R[1]) / S; q[1] = (R[2] + R[6]) / S; q[2] = (R[5] + R[7]) / S; q[3] = 0.25 * S;
  }
  if (q[0] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  double sn = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double th = 2.0 * atan2(sn, q[0]);
  double k = sn > 1e-150 ? th / sn : 2.0;
  rv[0] = k * q[1]; rv[1] = k * q[2]; rv[2] = k * q[3];
}

/* standing.jl:70-77: H = transform_to_root(pelvis), T = twist_wrt_world in the pelvis frame,
 * desired = pd(gains, rotation(H), T.angular), expressed in the pelvis frame */
void ora_standing_orientation_law(const ora_state *s, int body, double kp, double kd, double out[3]) {
  double R[9], p[3], T[6], rv[3];
  ora_get_transform(s, body, R, p);
  ora_get_twist(s, body, T);
  rotvec_from_R(R, rv);
  for (int a = 0; a < 3; a++) {
    double wb = R[0 + a] * T[0] + R[3 + a] * T[1] + R[6 + a] * T[2]; /* R' omega_world */
    out[a] = -kp * rv[a] - kd * wb;
  }
}

/* standing.jl:80-86: pd(gains, q, ref, v, 0) */
double ora_standing_joint_law(double kp, double kd, double q, double qref, double v) {
  return -kp * (q - qref) - kd * (v - 0.0);
}
