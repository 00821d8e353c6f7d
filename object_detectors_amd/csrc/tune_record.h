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

TuneMap& tune_timed(int id);       // shapes timed in THIS process (value unused): the 11 equal layers of a Darknet stage are timed once, not 11 times

// true: do not time this shape - its choice came from a locked record, or this process has timed it already (the choice stands either way)
inline bool tune_locked_has(int id, unsigned long long key) {
  const TuneMap& m = tune_table(id);
  if (m.find(key) == m.end()) return false;
  if (g_tune_locked) return true;
  const TuneMap& t = tune_timed(id);
  return t.find(key) != t.end();
}
inline void tune_mark_timed(int id, unsigned long long key) { tune_timed(id)[key] = 1; }

}  // namespace mi355
