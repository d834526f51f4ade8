// Tuning / ablation knobs (MF_TARGET_LANES, MF_KF_IMPL, MF_KF_X, MF_KF_DEBUG, MF_RED_*, MF_BTD_*, MF_BIG_*) exist only in
// builds made with -DMF_EXPERIMENT (`make EXTRA=-DMF_EXPERIMENT BUILD=build_exp OUT=../libmarkovflow_amd_exp.so`, selected at
// run time with MF_LIB_PATH, for A/B timing).  The shipped library reads no environment: nothing outside the arguments of a
// call can change its result.
#pragma once
#include <cstdlib>

namespace mf {
#ifdef MF_EXPERIMENT
inline const char* mf_knob(const char* name) { return std::getenv(name); }
#else
inline const char* mf_knob(const char*) { return nullptr; }
#endif
}   // namespace mf
