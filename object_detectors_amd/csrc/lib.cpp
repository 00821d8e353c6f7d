#include "common.h"
#include "tune_record.h"

#include <string.h>

#include <algorithm>
#include <vector>

namespace mi355 {
thread_local char g_err[512] = "";
bool g_tune_locked = false;
TuneMap& tune_table(int id) {
  static TuneMap tables[TUNE_TABLES];
  return tables[(unsigned)id < (unsigned)TUNE_TABLES ? id : 0];
}
TuneMap& tune_timed(int id) {
  static TuneMap tables[TUNE_TABLES];
  return tables[(unsigned)id < (unsigned)TUNE_TABLES ? id : 0];
}
}  // namespace mi355

using namespace mi355;

extern "C" {
const char* mi355det_last_error(void) { return mi355::g_err; }
int mi355det_version(void) { return 1; }

// ---- tune record (include/mi355det.h): entries of 16 bytes {u32 table, i32 value, u64 key}, sorted by (table, key) so that equal
//      records are equal byte strings
size_t mi355det_tune_export(void* buf, size_t cap) {
  std::vector<mi355det_tune_entry> v;
  for (int t = 0; t < TUNE_TABLES; ++t)
    for (const auto& kv : tune_table(t)) v.push_back(mi355det_tune_entry{(uint32_t)t, (int32_t)kv.second, (uint64_t)kv.first});
  std::sort(v.begin(), v.end(), [](const mi355det_tune_entry& a, const mi355det_tune_entry& b) { return a.table != b.table ? a.table < b.table : a.key < b.key; });
  const size_t need = v.size() * sizeof(mi355det_tune_entry);
  if (buf && cap >= need && need) memcpy(buf, v.data(), need);
  return need;
}

int mi355det_tune_import(const void* buf, size_t bytes, int replace) {
  if (bytes % sizeof(mi355det_tune_entry) != 0 || (bytes && !buf)) return fail(MI355DET_EINVAL, "%s: record size is not a whole number of entries", "tune_import");
  const mi355det_tune_entry* e = (const mi355det_tune_entry*)buf;
  const size_t n = bytes / sizeof(mi355det_tune_entry);
  for (size_t i = 0; i < n; ++i)
    if (e[i].table >= (uint32_t)TUNE_TABLES) return fail(MI355DET_EINVAL, "%s: unknown table id %lld", "tune_import", (long long)e[i].table);
  if (replace)
    for (int t = 0; t < TUNE_TABLES; ++t) {
      tune_table(t).clear();
      tune_timed(t).clear();
    }
  for (size_t i = 0; i < n; ++i) tune_table((int)e[i].table)[(unsigned long long)e[i].key] = (int)e[i].value;
  return MI355DET_OK;
}

int mi355det_tune_lock(int on) {
  const int was = g_tune_locked ? 1 : 0;
  g_tune_locked = on != 0;
  return was;
}

int mi355det_tune_clear(void) {
  for (int t = 0; t < TUNE_TABLES; ++t) {
    tune_table(t).clear();
    tune_timed(t).clear();
  }
  g_tune_locked = false;
  return MI355DET_OK;
}
}
