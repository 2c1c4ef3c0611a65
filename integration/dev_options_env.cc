// dev_options_env.cc -- TEST INFRASTRUCTURE, not part of the drop-in: linked only into the oracle/_ref/*_hip_seamtest binaries (the
// reference's own main.cc has no argv to spare), it copies GM_DEV_OPTIONS="NAME=VALUE,NAME=VALUE" from the environment into the
// library's developer options (gm_dev_option, include/graphminer_amd.h) before main() runs.  The shipped library, the shim
// (hip_solvers.cc) and the *_hip_base / *_hip_multigpu binaries read nothing from the environment.
#include <cstdlib>
#include <string>

#include "graphminer_amd.h"

namespace {
struct FromEnv {
  FromEnv() {
    const char *e = std::getenv("GM_DEV_OPTIONS");
    if (!e) return;
    std::string s(e);
    size_t i = 0;
    while (i < s.size()) {
      size_t j = s.find(',', i);
      if (j == std::string::npos) j = s.size();
      const std::string kv = s.substr(i, j - i);
      const size_t eq = kv.find('=');
      if (eq != std::string::npos && eq > 0) gm_dev_option(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str());
      i = j + 1;
    }
  }
} g_from_env;
}  // namespace
