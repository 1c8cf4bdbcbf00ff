/*
 * models.c -- built-in kinematic trees (plain C, no HIP).  See include/loik_amd_models.h.
 *
 * The reference obtains these from URDF files of example-robot-data 4.1.0 (pixi.lock:19) through
 * pinocchio::urdf::buildModel(urdf, model, false) (tests/loik-loid.cpp:110-111, :208-211): fixed base, fixed
 * joints merged into their parent, joints numbered depth-first.  No URDF exists in this image, so the tables
 * below restate the public robot descriptions by hand; they are "Panda" / "Talos-topology" for the purposes
 * of throughput and parity (same arithmetic per joint), not bit-comparable with a Pinocchio+URDF run.
 */
#include "loik_amd_models.h"

#include <math.h>
#include <string.h>

#define MAXJ 64

typedef struct {
  const char *name;
  int built;
  int nj, nq, nv;
  int parents[MAXJ], jtype[MAXJ], idx_q[MAXJ], idx_v[MAXJ];
  double axis[MAXJ * 3], placement[MAXJ * 12], q_lo[MAXJ], q_hi[MAXJ];
  const char *jname[MAXJ];
} table_t;

static table_t g_panda7 = {"panda7"}, g_panda9 = {"panda9"}, g_talos32 = {"talos32"}, g_talos32_ff = {"talos32_freeflyer"},
               g_talos44 = {"talos44"};

static void rpy_to_R(double r, double p, double y, double *R)
{
  /* URDF convention: R = Rz(yaw) Ry(pitch) Rx(roll); exact 0/+-1 for multiples of pi/2 */
  double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), cy = cos(y), sy = sin(y);
  double v[6] = {cr, sr, cp, sp, cy, sy};
  for (int i = 0; i < 6; ++i)
    if (fabs(v[i]) < 1e-15) v[i] = 0.0;
    else if (fabs(fabs(v[i]) - 1.0) < 1e-15) v[i] = v[i] > 0 ? 1.0 : -1.0;
  cr = v[0]; sr = v[1]; cp = v[2]; sp = v[3]; cy = v[4]; sy = v[5];
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

static void tbl_init(table_t *t)
{
  t->nj = 1; t->nq = 0; t->nv = 0;
  t->parents[0] = 0; t->jtype[0] = LOIKB_J_NONE; t->idx_q[0] = 0; t->idx_v[0] = 0;
  t->jname[0] = "universe";
  memset(t->axis, 0, sizeof(t->axis));
  memset(t->placement, 0, sizeof(t->placement));
  t->placement[0] = t->placement[4] = t->placement[8] = 1.0;
}

static int tbl_add(table_t *t, const char *name, int parent, int jtype, double ax, double ay, double az,
                   double x, double y, double z, double roll, double pitch, double yaw, double lo, double hi)
{
  int i = t->nj++;
  t->jname[i] = name;
  t->parents[i] = parent;
  t->jtype[i] = jtype;
  t->idx_q[i] = t->nq;
  t->idx_v[i] = t->nv;
  double *a = t->axis + 3 * i;
  switch (jtype) {
  case LOIKB_J_RX: case LOIKB_J_PX: a[0] = 1; break;
  case LOIKB_J_RY: case LOIKB_J_PY: a[1] = 1; break;
  case LOIKB_J_RZ: case LOIKB_J_PZ: a[2] = 1; break;
  default: a[0] = ax; a[1] = ay; a[2] = az; break;
  }
  rpy_to_R(roll, pitch, yaw, t->placement + 12 * i);
  t->placement[12 * i + 9] = x; t->placement[12 * i + 10] = y; t->placement[12 * i + 11] = z;
  if (jtype == LOIKB_J_FREEFLYER) {
    /* translation in [lo, hi]^3; the quaternion entries are placeholders (callers draw unit quaternions) */
    for (int k = 0; k < 3; ++k) { t->q_lo[t->nq + k] = lo; t->q_hi[t->nq + k] = hi; }
    for (int k = 3; k < 7; ++k) { t->q_lo[t->nq + k] = -1.0; t->q_hi[t->nq + k] = 1.0; }
    t->nq += 7; t->nv += 6;
  } else {
    t->q_lo[t->nq] = lo; t->q_hi[t->nq] = hi;
    t->nq += 1; t->nv += 1;
  }
  return i;
}

static void build_panda(table_t *t, int fingers)
{
  const double H = 1.57079632679489661923; /* pi/2 */
  tbl_init(t);
  int j = 0;
  j = tbl_add(t, "panda_joint1", j, LOIKB_J_RZ, 0, 0, 0, 0, 0, 0.333, 0, 0, 0, -2.8973, 2.8973);
  j = tbl_add(t, "panda_joint2", j, LOIKB_J_RZ, 0, 0, 0, 0, 0, 0, -H, 0, 0, -1.7628, 1.7628);
  j = tbl_add(t, "panda_joint3", j, LOIKB_J_RZ, 0, 0, 0, 0, -0.316, 0, H, 0, 0, -2.8973, 2.8973);
  j = tbl_add(t, "panda_joint4", j, LOIKB_J_RZ, 0, 0, 0, 0.0825, 0, 0, H, 0, 0, -3.0718, -0.0698);
  j = tbl_add(t, "panda_joint5", j, LOIKB_J_RZ, 0, 0, 0, -0.0825, 0.384, 0, -H, 0, 0, -2.8973, 2.8973);
  j = tbl_add(t, "panda_joint6", j, LOIKB_J_RZ, 0, 0, 0, 0, 0, 0, H, 0, 0, -0.0175, 3.7525);
  j = tbl_add(t, "panda_joint7", j, LOIKB_J_RZ, 0, 0, 0, 0.088, 0, 0, H, 0, 0, -2.8973, 2.8973);
  if (fingers) {
    /* panda_joint8 (fixed, z 0.107) + panda_hand_joint (fixed, yaw -pi/4) merged into the finger placements */
    tbl_add(t, "panda_finger_joint1", j, LOIKB_J_PY, 0, 0, 0, 0, 0, 0.107 + 0.0584, 0, 0, -H / 2, 0.0, 0.04);
    tbl_add(t, "panda_finger_joint2", j, LOIKB_J_PU, 0, -1, 0, 0, 0, 0.107 + 0.0584, 0, 0, -H / 2, 0.0, 0.04);
  }
  t->built = 1;
}

/* floating = 1: Pinocchio's buildModel(urdf, JointModelFreeFlyer(), model): joint 1 = "root_joint" (free-flyer,
 * identity placement), the robot's root children hang off it */
/* The six passive finger joints under a wrist of talos_full_v2.urdf (the file the reference's fixture loads,
 * tests/loik-loid.cpp:110-111; its <mimic> tags are ignored by pinocchio::urdf::buildModel, so each is a revolute DoF):
 * the wrist link then carries FOUR joints (gripper, inner_double, inner_single, motor_single), inner_double two fingertips,
 * inner_single one.  Names and topology as in the public talos_data gripper description; the centimetre-scale placements
 * are representative values, not the URDF's (no URDF in this image) -- same arithmetic per joint, a synthetic geometry. */
static void add_fingers(table_t *t, int wrist, const char *const *nm, double sy)
{
  int j = tbl_add(t, nm[0], wrist, LOIKB_J_RX, 0, 0, 0, 0, sy * 0.018, -0.135, 0, 0, 0, -0.3, 0.3);   /* inner_double */
  tbl_add(t, nm[1], j, LOIKB_J_RX, 0, 0, 0, 0.032, sy * 0.004, -0.063, 0, 0, 0, -0.3, 0.3);           /* fingertip_1  */
  tbl_add(t, nm[2], j, LOIKB_J_RX, 0, 0, 0, -0.032, sy * 0.004, -0.063, 0, 0, 0, -0.3, 0.3);          /* fingertip_2  */
  j = tbl_add(t, nm[3], wrist, LOIKB_J_RX, 0, 0, 0, 0, sy * -0.018, -0.135, 0, 0, 0, -0.3, 0.3);      /* inner_single */
  tbl_add(t, nm[4], j, LOIKB_J_RX, 0, 0, 0, 0, sy * -0.004, -0.063, 0, 0, 0, -0.3, 0.3);              /* fingertip_3  */
  tbl_add(t, nm[5], wrist, LOIKB_J_RX, 0, 0, 0, 0, sy * -0.02025, -0.12193, 0, 0, 0, -0.3, 0.3);      /* motor_single */
}

static void build_talos32(table_t *t, int floating, int fingers)
{
  static const char *const fl[6] = {"gripper_left_inner_double_joint", "gripper_left_fingertip_1_joint",
                                    "gripper_left_fingertip_2_joint", "gripper_left_inner_single_joint",
                                    "gripper_left_fingertip_3_joint", "gripper_left_motor_single_joint"};
  static const char *const fr[6] = {"gripper_right_inner_double_joint", "gripper_right_fingertip_1_joint",
                                    "gripper_right_fingertip_2_joint", "gripper_right_inner_single_joint",
                                    "gripper_right_fingertip_3_joint", "gripper_right_motor_single_joint"};
  tbl_init(t);
  const double d = 0.3; /* sampling half-range around the nominal pose */
  int j;
  const int base = floating ? tbl_add(t, "root_joint", 0, LOIKB_J_FREEFLYER, 0, 0, 0, 0, 0, 0, 0, 0, 0, -0.5, 0.5) : 0;
  /* left leg (root child) */
  j = tbl_add(t, "leg_left_1_joint", base, LOIKB_J_RZ, 0, 0, 0, -0.02, 0.085, -0.27105, 0, 0, 0, -d, d);
  j = tbl_add(t, "leg_left_2_joint", j, LOIKB_J_RX, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  j = tbl_add(t, "leg_left_3_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, 0, 0, 0, 0, -0.4 - d, -0.4 + d);
  j = tbl_add(t, "leg_left_4_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, -0.38, 0, 0, 0, 0.8 - d, 0.8 + d);
  j = tbl_add(t, "leg_left_5_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, -0.325, 0, 0, 0, -0.4 - d, -0.4 + d);
  j = tbl_add(t, "leg_left_6_joint", j, LOIKB_J_RX, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  /* right leg (root child) */
  j = tbl_add(t, "leg_right_1_joint", base, LOIKB_J_RZ, 0, 0, 0, -0.02, -0.085, -0.27105, 0, 0, 0, -d, d);
  j = tbl_add(t, "leg_right_2_joint", j, LOIKB_J_RX, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  j = tbl_add(t, "leg_right_3_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, 0, 0, 0, 0, -0.4 - d, -0.4 + d);
  j = tbl_add(t, "leg_right_4_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, -0.38, 0, 0, 0, 0.8 - d, 0.8 + d);
  j = tbl_add(t, "leg_right_5_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, -0.325, 0, 0, 0, -0.4 - d, -0.4 + d);
  j = tbl_add(t, "leg_right_6_joint", j, LOIKB_J_RX, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  /* torso (root child) */
  j = tbl_add(t, "torso_1_joint", base, LOIKB_J_RZ, 0, 0, 0, 0, 0, 0.0722, 0, 0, 0, -d, d);
  int torso2 = tbl_add(t, "torso_2_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  /* left arm */
  j = tbl_add(t, "arm_left_1_joint", torso2, LOIKB_J_RZ, 0, 0, 0, 0, 0.1575, 0.232, 0, 0, 0, 0.25 - d, 0.25 + d);
  j = tbl_add(t, "arm_left_2_joint", j, LOIKB_J_RX, 0, 0, 0, 0.00493378, 0.1365, 0.04673, 0, 0, 0, 0.17 - d, 0.17 + d);
  j = tbl_add(t, "arm_left_3_joint", j, LOIKB_J_RZ, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  j = tbl_add(t, "arm_left_4_joint", j, LOIKB_J_RY, 0, 0, 0, 0.02, 0, -0.273, 0, 0, 0, -0.6 - d, -0.6 + d);
  j = tbl_add(t, "arm_left_5_joint", j, LOIKB_J_RZ, 0, 0, 0, -0.02, 0, -0.2643, 0, 0, 0, -d, d);
  j = tbl_add(t, "arm_left_6_joint", j, LOIKB_J_RX, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  j = tbl_add(t, "arm_left_7_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  tbl_add(t, "gripper_left_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0.02025, -0.12193, 0, 0, 0, -0.5, 0.0);
  if (fingers) add_fingers(t, j, fl, 1.0);
  /* right arm */
  j = tbl_add(t, "arm_right_1_joint", torso2, LOIKB_J_RZ, 0, 0, 0, 0, -0.1575, 0.232, 0, 0, 0, -0.25 - d, -0.25 + d);
  j = tbl_add(t, "arm_right_2_joint", j, LOIKB_J_RX, 0, 0, 0, 0.00493378, -0.1365, 0.04673, 0, 0, 0, -0.17 - d, -0.17 + d);
  j = tbl_add(t, "arm_right_3_joint", j, LOIKB_J_RZ, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  j = tbl_add(t, "arm_right_4_joint", j, LOIKB_J_RY, 0, 0, 0, 0.02, 0, -0.273, 0, 0, 0, -0.6 - d, -0.6 + d);
  j = tbl_add(t, "arm_right_5_joint", j, LOIKB_J_RZ, 0, 0, 0, -0.02, 0, -0.2643, 0, 0, 0, -d, d);
  j = tbl_add(t, "arm_right_6_joint", j, LOIKB_J_RX, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  j = tbl_add(t, "arm_right_7_joint", j, LOIKB_J_RY, 0, 0, 0, 0, 0, 0, 0, 0, 0, -d, d);
  tbl_add(t, "gripper_right_joint", j, LOIKB_J_RY, 0, 0, 0, 0, -0.02025, -0.12193, 0, 0, 0, -0.5, 0.0);
  if (fingers) add_fingers(t, j, fr, -1.0);
  /* head */
  j = tbl_add(t, "head_1_joint", torso2, LOIKB_J_RY, 0, 0, 0, 0, 0, 0.316, 0, 0, 0, -d, d);
  j = tbl_add(t, "head_2_joint", j, LOIKB_J_RZ, 0, 0, 0, 0.02, 0, 0, 0, 0, 0, -d, d);
  t->built = 1;
}

static table_t *find(const char *name)
{
  if (!name) return 0;
  if (!strcmp(name, "panda7")) { if (!g_panda7.built) build_panda(&g_panda7, 0); return &g_panda7; }
  if (!strcmp(name, "panda9")) { if (!g_panda9.built) build_panda(&g_panda9, 1); return &g_panda9; }
  if (!strcmp(name, "talos32")) { if (!g_talos32.built) build_talos32(&g_talos32, 0, 0); return &g_talos32; }
  if (!strcmp(name, "talos32_freeflyer")) { if (!g_talos32_ff.built) build_talos32(&g_talos32_ff, 1, 0); return &g_talos32_ff; }
  if (!strcmp(name, "talos44")) { if (!g_talos44.built) build_talos32(&g_talos44, 0, 1); return &g_talos44; }
  return 0;
}

int loikb_builtin_model(const char *name, loikb_model_desc *out, const double **q_lo, const double **q_hi)
{
  table_t *t = find(name);
  if (!t || !out) return -1;
  out->njoints = t->nj;
  out->nq = t->nq;
  out->nv = t->nv;
  out->parents = t->parents;
  out->jtype = t->jtype;
  out->axis = t->axis;
  out->idx_q = t->idx_q;
  out->idx_v = t->idx_v;
  out->placement = t->placement;
  out->comp_first = 0; out->comp_count = 0; out->comp_jtype = 0; out->comp_axis = 0; out->comp_placement = 0; out->pitch = 0; out->comp_pitch = 0;
  if (q_lo) *q_lo = t->q_lo;
  if (q_hi) *q_hi = t->q_hi;
  return 0;
}

const char *loikb_builtin_joint_name(const char *name, int joint)
{
  table_t *t = find(name);
  if (!t || joint < 0 || joint >= t->nj) return 0;
  return t->jname[joint];
}

int loikb_builtin_joint_id(const char *name, const char *joint_name)
{
  table_t *t = find(name);
  if (!t || !joint_name) return -1;
  for (int i = 0; i < t->nj; ++i)
    if (!strcmp(t->jname[i], joint_name)) return i;
  return -1;
}
