// Tiny gflags-compatible command line parser for the headless drivers (gflags is not installed): --name=value,
// --name value, --name / --noname for booleans; same flag names and defaults as the reference's DEFINE_* lines.
#pragma once
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>

namespace mvicp {
class Flags {
 public:
  Flags(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
      std::string a(argv[i]);
      if (a.compare(0, 2, "--") == 0) a = a.substr(2);
      else if (a.compare(0, 1, "-") == 0) a = a.substr(1);
      else continue;
      const size_t eq = a.find('=');
      if (eq != std::string::npos) { kv_[a.substr(0, eq)] = a.substr(eq + 1); continue; }
      if (i + 1 < argc && argv[i + 1][0] != '-') { kv_[a] = argv[++i]; continue; }
      if (a.compare(0, 2, "no") == 0 && a.size() > 2) kv_[a.substr(2)] = "false";
      else kv_[a] = "true";
    }
  }
  bool b(const std::string& n, bool d) const { auto it = kv_.find(n); if (it == kv_.end()) return d; return !(it->second == "false" || it->second == "0"); }
  int i(const std::string& n, int d) const { auto it = kv_.find(n); return it == kv_.end() ? d : std::atoi(it->second.c_str()); }
  double f(const std::string& n, double d) const { auto it = kv_.find(n); return it == kv_.end() ? d : std::atof(it->second.c_str()); }
  std::string s(const std::string& n, const std::string& d) const { auto it = kv_.find(n); return it == kv_.end() ? d : it->second; }
 private:
  std::map<std::string, std::string> kv_;
};
}  // namespace mvicp
