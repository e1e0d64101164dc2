// TEST INFRASTRUCTURE ONLY (oracle/_ref): C wrapper that restates my_cpp get_ik_within_limits
// (my_cpp/common.cpp:9-72) over the reference's vendored IKFast solver for the KUKA iiwa14.  The solver source is
// compiled from where it lies under /root/reference by oracle/build_ref.py; it is never copied into this repository.
#define IKFAST_HAS_LIBRARY
#ifndef IKFAST_NO_MAIN
#define IKFAST_NO_MAIN
#endif
#include "ikfast.h"
#include <vector>

using namespace ikfast;

// 1 iff some IK solution of ee_in_base (row-major 4x4 float) lies within [lower, upper] (7 joints), free joint = 0
extern "C" int ik_within_limits(const float* ee16, const double* upper, const double* lower) {
  IkSolutionList<double> solutions;
  std::vector<double> vfree(GetNumFreeParameters());           // value-initialised to 0 (common.cpp:15)
  double eerot[9], eetrans[3];
  for (int i = 0; i < 3; ++i) eetrans[i] = ee16[i * 4 + 3];
  for (int h = 0; h < 3; ++h) for (int w = 0; w < 3; ++w) eerot[h * 3 + w] = ee16[h * 4 + w];
  if (!ComputeIk(eetrans, eerot, &vfree[0], solutions)) return 0;
  std::vector<double> sol(GetNumJoints());
  for (std::size_t i = 0; i < solutions.GetNumSolutions(); ++i) {
    const IkSolutionBase<double>& s = solutions.GetSolution(i);
    std::vector<double> vsolfree(s.GetFree().size());
    s.GetSolution(&sol[0], vsolfree.size() > 0 ? &vsolfree[0] : NULL);
    bool bad = false;
    for (std::size_t j = 0; j < sol.size(); ++j) if (sol[j] > upper[j] || sol[j] < lower[j]) { bad = true; break; }
    if (!bad) return 1;
  }
  return 0;
}

extern "C" void ik_fk(const double* joints, double* eetrans, double* eerot) { ComputeFk(joints, eetrans, eerot); }

// all solutions (free values = 0 as in common.cpp:46-49): writes up to max_solutions x 7 joint values, returns their number
extern "C" int ik_solutions(const float* ee16, double* out, int max_solutions) {
  IkSolutionList<double> solutions;
  std::vector<double> vfree(GetNumFreeParameters());
  double eerot[9], eetrans[3];
  for (int i = 0; i < 3; ++i) eetrans[i] = ee16[i * 4 + 3];
  for (int h = 0; h < 3; ++h) for (int w = 0; w < 3; ++w) eerot[h * 3 + w] = ee16[h * 4 + w];
  if (!ComputeIk(eetrans, eerot, &vfree[0], solutions)) return 0;
  std::vector<double> sol(GetNumJoints());
  int n = 0;
  for (std::size_t i = 0; i < solutions.GetNumSolutions() && n < max_solutions; ++i, ++n) {
    const IkSolutionBase<double>& s = solutions.GetSolution(i);
    std::vector<double> vsolfree(s.GetFree().size());
    s.GetSolution(&sol[0], vsolfree.size() > 0 ? &vsolfree[0] : NULL);
    for (int j = 0; j < 7; ++j) out[n * 7 + j] = sol[j];
  }
  return n;
}
