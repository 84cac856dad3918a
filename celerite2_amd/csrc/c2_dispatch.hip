// c2_dispatch.hip -- storage and C-ABI of the option table (c2_dispatch.hpp).  The environment is read once, by the
// static initialiser below, when the library is loaded.
#include "c2_dispatch.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/celerite2_amd.h"

namespace c2 {
namespace opt {
namespace {
struct Entry {
  const char *name, *env;
  double def;
  char kind;
  const char *doc, *src;
};
const Entry kTable[kCount] = {
#define C2_OPT_ROW(id, env, def, kind, doc, src) {#id, env, (double)(def), kind, doc, src},
    C2_OPTIONS(C2_OPT_ROW)
#undef C2_OPT_ROW
};
std::atomic<double> g_val[kCount];
std::atomic<bool> g_set[kCount];

// A value counts only if strtod consumes all of it (trailing blanks allowed): "C2_MFMA=on" or a threshold of "abc" leaves
// the option UNSET -- the same rule c2_set_option applies -- instead of silently becoming 0.
bool parse(const char *e, double *v) {
  if (!e || !*e) return false;
  char *end = nullptr;
  const double x = strtod(e, &end);
  if (end == e) return false;
  while (*end == ' ' || *end == '\t' || *end == '\n') ++end;
  if (*end) return false;
  *v = x;
  return true;
}
void load_env() {
  for (int i = 0; i < kCount; ++i) {
    double v;
    const char *e = getenv(kTable[i].env);
    if (parse(e, &v)) {
      g_val[i].store(v);
      g_set[i].store(true);
    } else {
      if (e && *e) fprintf(stderr, "celerite2_amd: ignoring unparsable %s=%s\n", kTable[i].env, e);
      g_val[i].store(kTable[i].def);
      g_set[i].store(false);
    }
  }
}
struct Init {
  Init() { load_env(); }
} g_init;

int find(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < kCount; ++i)
    if (!strcmp(name, kTable[i].name) || !strcmp(name, kTable[i].env)) return i;
  return -1;
}
}  // namespace

bool has(Id id) { return g_set[id].load(std::memory_order_relaxed); }
double val(Id id) { return g_val[id].load(std::memory_order_relaxed); }
}  // namespace opt
}  // namespace c2

using namespace c2::opt;

extern "C" {

int c2_set_option(const char *name, const char *value) {
  const int i = find(name);
  if (i < 0) return C2_ERR_INVALID;
  if (value && *value) {
    double v;
    if (!parse(value, &v)) return C2_ERR_INVALID;
    g_val[i].store(v);
    g_set[i].store(true);
  } else {   // back to the table's default / the automatic choice
    g_val[i].store(kTable[i].def);
    g_set[i].store(false);
  }
  return C2_OK;
}

int c2_get_option(const char *name, double *value, int *is_set) {
  const int i = find(name);
  if (i < 0) return C2_ERR_INVALID;
  if (value) *value = g_val[i].load();
  if (is_set) *is_set = g_set[i].load() ? 1 : 0;
  return C2_OK;
}

int c2_option_count(void) { return kCount; }

int c2_option_info(int index, const char **name, const char **env, double *default_value, int *is_switch,
                   const char **doc, const char **measured) {
  if (index < 0 || index >= kCount) return C2_ERR_INVALID;
  const Entry &e = kTable[index];
  if (name) *name = e.name;
  if (env) *env = e.env;
  if (default_value) *default_value = e.def;
  if (is_switch) *is_switch = e.kind == 's';
  if (doc) *doc = e.doc;
  if (measured) *measured = e.src;
  return C2_OK;
}

void c2_options_reload_env(void) { load_env(); }

}  // extern "C"
