// The tune record: every choice the plan build makes by TIMING candidates (tile configuration per implicit-GEMM shape, the form of the
// stride-2 data gradient, the split count of the weight gradient).  The choices fix the summation order of the kernels, so two boxes that
// time differently compute (slightly) different numbers; exporting the record on one box / rank and importing it LOCKED on another makes the
// kernels, their speed and their summation order identical (mi355det_tune_export / _import / _lock; host mirror: object_detectors_amd/tune.py).
#pragma once
#include <unordered_map>

namespace mi355 {

enum { TUNE_IGEMM = 0, TUNE_S2CAT = 1, TUNE_WGRAD = 2, TUNE_TABLES = 3 };

typedef std::unordered_map<unsigned long long, int> TuneMap;
TuneMap& tune_table(int id);       // defined in lib.cpp (one copy for the whole library)
extern bool g_tune_locked;         // true: a shape that has an entry is never timed again (the entry came from a record)

inline bool tune_locked_has(int id, unsigned long long key) {
  if (!g_tune_locked) return false;
  const TuneMap& m = tune_table(id);
  return m.find(key) != m.end();
}

}  // namespace mi355
