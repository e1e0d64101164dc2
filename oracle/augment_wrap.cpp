// TEST INFRASTRUCTURE ONLY.  C entry point around the REFERENCE's own directionVecToRotation / augmentGraspPoses
// (my_cpp/common.cpp:75-153).  my_cpp as a whole cannot be built here (FCL / octomap / boost absent), but these two functions
// depend on Eigen alone, and Eigen is vendored in the reference tree (PointGroup/lib/pointgroup_ops/eigen3).  oracle/build_ref.py
// extracts exactly those source lines from /root/reference AT BUILD TIME into oracle/_ref/augment_extract.inc (git-ignored,
// never committed) and compiles this file around them; nothing of the reference's text lives in the repository.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>
#include <Eigen/Dense>
#include <Eigen/Geometry>
using namespace Eigen;
using vectorMatrix4f = std::vector<Eigen::Matrix4f, Eigen::aligned_allocator<Eigen::Matrix4f>>;   // my_cpp/common.h:51

#include "_ref/augment_extract.inc"

extern "C" int ref_direction_vec_to_rotation(const float* direction, const float* ref, float* out9) {
  Eigen::Matrix3f R = directionVecToRotation(Eigen::Vector3f(direction[0], direction[1], direction[2]), Eigen::Vector3f(ref[0], ref[1], ref[2]));
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out9[r * 3 + c] = R(r, c);
  return 0;
}

// sphere_pts: (S,3) row-major.  Returns the number of poses the reference produced; writes at most `cap` of them (row-major 4x4).
// NOTE the reference iterates i < sphere_pts.size() (= 3*S) and so reads S..3S-1 "rows" beyond the matrix; callers compare only
// the poses generated from the S valid rows, which come first.
extern "C" int ref_augment_grasp_poses(const float* R0, const float* p, const float* sphere_pts, int S, float inplane_rot_step,
                                       float hand_depth, float approach_step, float init_bite, float* out, int cap) {
  Eigen::Matrix3f R;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = R0[r * 3 + c];
  Eigen::MatrixXf sp(S, 3);
  for (int i = 0; i < S; ++i) for (int c = 0; c < 3; ++c) sp(i, c) = sphere_pts[i * 3 + c];
  vectorMatrix4f poses = augmentGraspPoses(R, Eigen::Vector3f(p[0], p[1], p[2]), sp, inplane_rot_step, hand_depth, approach_step, init_bite);
  const int n = (int)poses.size();
  for (int i = 0; i < n && i < cap; ++i)
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[i * 16 + r * 4 + c] = poses[i](r, c);
  return n;
}
