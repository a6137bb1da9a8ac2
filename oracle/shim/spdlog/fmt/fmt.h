// oracle/shim — minimal fmt::format used only to build exception messages in the reference TUs.
#pragma once
#include <string>
namespace fmt {
template <class... A> inline std::string format(const char* f, A&&...) { return std::string(f); }
} // namespace fmt
