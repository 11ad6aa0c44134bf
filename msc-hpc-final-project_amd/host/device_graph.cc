#include "device_graph.h"

#include <cstdlib>
#include <mutex>
#include <sstream>
#include <string>

#include "lzx.h"

deviceGraph::~deviceGraph() {
  if (owner && evict) evict(owner);
  for (lzx_ctx *h : ranks)
    if (h) lzx_destroy(h);
}

namespace {
std::mutex g_lock;
std::vector<int> g_override;
}  // namespace

void lzx_host_set_devices(const std::vector<int> &ids) {
  std::lock_guard<std::mutex> g(g_lock);
  g_override = ids;
}

std::vector<int> lzx_host_devices() {
  int count = 0;
  if (lzx_device_count(&count) != LZX_OK || count <= 0) return {};
  {
    std::lock_guard<std::mutex> g(g_lock);
    if (!g_override.empty()) return g_override;
  }
  const char *env = std::getenv("LZX_DEVICES");
  std::vector<int> ids;
  if (env && *env) {
    const std::string s(env);
    if (s == "all") {
      for (int i = 0; i < count; ++i) ids.push_back(i);
    } else if (s.find(',') == std::string::npos) {
      const int n = std::atoi(s.c_str());
      for (int i = 0; i < n; ++i) ids.push_back(i % count);
    } else {
      std::stringstream ss(s);
      std::string tok;
      while (std::getline(ss, tok, ','))
        if (!tok.empty()) ids.push_back(std::atoi(tok.c_str()) % count);
    }
  }
  if (ids.empty()) ids.push_back(0);
  return ids;
}
