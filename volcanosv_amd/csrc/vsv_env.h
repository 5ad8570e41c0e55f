// vsv_env.h — the library's test hooks and timing switches (VSV_SORT, VSV_PAIR, VSV_BK_CAP, VSV_CLR, VSV_SPLIT_STREAM, VSV_K1_*,
// VSV_BAM_WINDOW, VSV_TRACE_COUNTERS, ...) are honoured only when VSV_DEBUG=1 is set as well: results never depend on them, timing
// does, and a user's environment must not be able to flip them silently. tests/conftest.py sets VSV_DEBUG=1; the drivers do not.
#pragma once
#include <stdlib.h>
inline const char* vsv_dbg_env(const char* name) {
  const char* d = getenv("VSV_DEBUG");
  return (d && d[0] == '1') ? getenv(name) : nullptr;
}
