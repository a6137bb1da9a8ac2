// oracle/shim — no-op stand-in for the spdlog headers the reference TUs include, so that
// src/crypto/sha256_hasher.cpp and src/chunking/{rabin,streaming}_chunker.cpp compile unmodified
// out of /root/reference (spdlog itself is not installed here).  Test infrastructure only.
#pragma once
namespace spdlog {
template <class... A> inline void trace(A&&...) {}
template <class... A> inline void debug(A&&...) {}
template <class... A> inline void info(A&&...) {}
template <class... A> inline void warn(A&&...) {}
template <class... A> inline void error(A&&...) {}
template <class... A> inline void critical(A&&...) {}
} // namespace spdlog
