/*
 * loik_amd_models.h -- kinematic-tree description handed across the C-ABI, plus built-in robot tables.
 *
 * The reference takes a `pinocchio::ModelTpl<double>` by const-ref in the solver constructor
 * (include/loik/loik-loid-optimized.hpp:129-134) and reads exactly these members on the hot path:
 * `njoints`, `nv`, `parents[]`, `joints[i]` (type, `idx_q()`, `idx_v()`, `nv()==1`), `jointPlacements[i]`
 * (loik-loid-optimized.hxx:46-47, :118-119, :257-265).  `loikb_model_desc` carries the same data with the
 * same names and memory order (joint 0 = universe, parents[i] < i, 6-vectors [linear; angular]) so a
 * Pinocchio -> loikb adapter is a field-by-field copy.
 *
 * Joints: all of Pinocchio's 1-DoF types (incl. the unbounded revolute joints with q = (cos, sin)), the multi-DoF types
 * whose motion subspace is a constant selection of the columns of I6 -- free-flyer (floating base), spherical, translation,
 * planar --, JointModelSphericalZYX, whose q-dependent subspace is that of a Z-Y-X revolute chain, and
 * JointModelComposite of any of those (LOIKB_J_COMPOSITE + the comp_* arrays below; up to 6 DoF -- e.g. the hand-made floating
 * base "translation + spherical").  JointModelUniversal(axis1, axis2) is the composite of RevoluteUnaligned(axis1) and
 * RevoluteUnaligned(axis2) with identity placements -- pass it as that (the Pinocchio adapter does).  The helical joints (S = [pitch a; a]) are 1-DoF joints with a pitch
 * (LOIKB_J_HX .. HU + the pitch array).  Not covered: JointModelMimic (two joints share one coordinate: the elimination couples them
 * across the tree).  On the device a multi-DoF
 * joint is a chain of 1-DoF joints with massless links in between; the caller never sees that: q, z / nu / w / lb / ub
 * (length model.nv, Pinocchio's idx_v order) and the per-link results are the caller's model's.
 */
#ifndef LOIK_AMD_MODELS_H
#define LOIK_AMD_MODELS_H

#ifdef __cplusplus
extern "C" {
#endif

/* joint types; names follow Pinocchio's JointModel{RX,RY,RZ,PX,PY,PZ,RevoluteUnaligned,PrismaticUnaligned,
 * FreeFlyer,Spherical,Translation,SphericalZYX,Planar,RUBX/Y/Z,Composite} */
enum {
  LOIKB_J_NONE = 0, /* universe */
  LOIKB_J_RX = 1,
  LOIKB_J_RY = 2,
  LOIKB_J_RZ = 3,
  LOIKB_J_PX = 4,
  LOIKB_J_PY = 5,
  LOIKB_J_PZ = 6,
  LOIKB_J_RU = 7,
  LOIKB_J_PU = 8,
  LOIKB_J_FREEFLYER = 9,   /* nq 7: translation, quaternion (x,y,z,w); nv 6: [linear; angular] in the joint frame */
  LOIKB_J_SPHERICAL = 10,  /* nq 4: quaternion (x,y,z,w);              nv 3: angular velocity in the joint frame  */
  LOIKB_J_TRANSLATION = 11,/* nq 3, nv 3                                                                          */
  LOIKB_J_SPHERICAL_ZYX = 12, /* JointModelSphericalZYX: nq 3 (angles about z, y, x: R = Rz Ry Rx), nv 3 = their rates.
                                 The motion subspace depends on q; on the device it is the chain RZ -> RY -> RX it describes */
  LOIKB_J_PLANAR = 13,     /* JointModelPlanar: nq 4 (x, y, cos, sin), nv 3 (vx, vy, wz in the joint frame)               */
  LOIKB_J_RUBX = 14,       /* JointModelRUBX / RUBY / RUBZ (revolute unbounded): nq 2 (cos, sin), nv 1                     */
  LOIKB_J_RUBY = 15,
  LOIKB_J_RUBZ = 16,
  LOIKB_J_COMPOSITE = 17   /* JointModelComposite: see loikb_model_desc.comp_*; nq / nv = the sums over its sub-joints (any type
                              above or RUBU, not a composite; nv <= 6), coordinates in sub-joint order.  M = prod_k (placement_k *
                              M_k(q_k)), the motion subspace columns of sub-joint k are its S_k seen from the last sub-joint's
                              frame (q-dependent); on the device it is the chain of its sub-joints (each multi-DoF one its own
                              chain) with massless links in between                                                          */
  ,
  LOIKB_J_RUBU = 18,       /* JointModelRevoluteUnboundedUnaligned: nq 2 (cos, sin), nv 1, about `axis` (also as a sub-joint of a
                              composite)                                                                                    */
  LOIKB_J_HX = 19,         /* JointModelHelicalX / Y / Z (JointModelHX ...) and JointModelHelicalUnaligned: nq 1, nv 1; a rotation by q  */
  LOIKB_J_HY = 20,         /* about the axis together with a translation of pitch * q along it: M(q) = (Rot(axis, q), pitch q axis),   */
  LOIKB_J_HZ = 21,         /* S = [pitch axis; axis].  The pitch of joint i is loikb_model_desc.pitch[i]; HU takes its axis from `axis`. */
  LOIKB_J_HU = 22          /* (As a sub-joint of a composite: comp_jtype / comp_axis / comp_pitch.)                                  */
};

typedef struct loikb_model_desc {
  int njoints;             /* model.njoints, includes the universe joint 0                    */
  int nq, nv;              /* model.nq, model.nv (== njoints-1 for all-1-DoF trees)           */
  const int *parents;      /* [njoints] model.parents                                         */
  const int *jtype;        /* [njoints] LOIKB_J_*                                             */
  const double *axis;      /* [njoints][3] unit axis in the joint frame (e_k for aligned)     */
  const int *idx_q;        /* [njoints] joints[i].idx_q(): cumulative in joint order          */
  const int *idx_v;        /* [njoints] joints[i].idx_v(): cumulative in joint order          */
  const double *placement; /* [njoints][12] jointPlacements[i]: R row-major (9), then t (3)   */
  /* JointModelComposite (all NULL / ignored when the model has none): joint i of type LOIKB_J_COMPOSITE consists of the
     sub-joints comp_first[i] .. comp_first[i] + comp_count[i] - 1 of the three arrays below -- what
     JointModelComposite::addJoint(jmodel, placement) stores: the sub-joint's type (any LOIKB_J_* but COMPOSITE), its
     axis (RU / PU / RUBU) and its placement relative to the previous sub-joint's frame (the first: relative to the frame
     jointPlacements[i] defines).  comp_count[i] <= 6 and at most 6 degrees of freedom per composite. */
  const int *comp_first;        /* [njoints] */
  const int *comp_count;        /* [njoints] */
  const int *comp_jtype;        /* [n_sub]   */
  const double *comp_axis;      /* [n_sub][3]  */
  const double *comp_placement; /* [n_sub][12] */
  const double *pitch;          /* [njoints] JointModelHelical*::m_pitch (read for LOIKB_J_HX .. HU only); NULL when the model has none */
  const double *comp_pitch;     /* [n_sub]   the same for helical SUB-joints of composites; NULL when there is none                    */
} loikb_model_desc;

/*
 * Built-in tables (restated from public URDF knowledge; not verifiable offline -- SURVEY.md Appendix C):
 *   "panda7"  : Franka Panda arm, 7 revolute-z joints, chain
 *   "panda9"  : panda7 + two prismatic fingers (PY and prismatic-unaligned -y), as example-robot-data's panda.urdf
 *   "talos32" : Talos humanoid topology, fixed base, legs(6+6) torso(2) arms(7+1, 7+1) head(2)
 *   "talos32_freeflyer" : the same robot under a free-flyer "root_joint" (floating base): 33 joints, nq 39, nv 38
 *   "talos44" : the topology of talos_full_v2.urdf, the file the reference's fixture loads (tests/loik-loid.cpp:110-111):
 *               talos32 + six passive finger joints under each wrist (nq = nv = 44, last joint head_2_joint); the wrist
 *               link carries four joints.  Finger placements are representative, not the URDF's
 * Returns 0 and fills *out with pointers to static storage valid for the process lifetime, or -1.
 * q_lo / q_hi (may be NULL) receive pointers to [nq] sampling ranges for synthetic configurations (the quaternion
 * entries of a free-flyer are placeholders: draw unit quaternions).
 */
int loikb_builtin_model(const char *name, loikb_model_desc *out, const double **q_lo, const double **q_hi);
/* joint name of a built-in model (index 0 = "universe"), or NULL */
const char *loikb_builtin_joint_name(const char *name, int joint);
/* index of a named joint in a built-in model, or -1 */
int loikb_builtin_joint_id(const char *name, const char *joint_name);

#ifdef __cplusplus
}
#endif
#endif
