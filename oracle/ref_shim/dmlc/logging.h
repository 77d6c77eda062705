/* Shim for dmlc-core's <dmlc/logging.h>, which is an EMPTY git submodule in the reference
 * checkout (third_party/dmlc-core).  Written from scratch: only the glog-style macros the
 * reference's CPU g-SpMM / g-SDDMM headers use.  A failed CHECK / LOG(FATAL) throws
 * dmlc::Error, as dmlc-core does when DMLC_LOG_FATAL_THROW is set (DGL's default).
 * Test infrastructure only (oracle/_ref). */
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>

namespace dmlc {
struct Error : public std::runtime_error {
  explicit Error(const std::string& s) : std::runtime_error(s) {}
};

class LogMessageFatal {
 public:
  LogMessageFatal(const char* file, int line) { os_ << file << ":" << line << ": "; }
  std::ostringstream& stream() { return os_; }
  ~LogMessageFatal() noexcept(false) { throw Error(os_.str()); }

 private:
  std::ostringstream os_;
};

class LogMessage {
 public:
  LogMessage(const char* file, int line) { os_ << file << ":" << line << ": "; }
  std::ostringstream& stream() { return os_; }
  ~LogMessage() { std::cerr << os_.str() << std::endl; }

 private:
  std::ostringstream os_;
};

// swallows the stream expression in `cond ? (void)0 : Voidify() & stream`
struct LogMessageVoidify {
  void operator&(std::ostream&) {}
};
}  // namespace dmlc

#define LOG_INFO ::dmlc::LogMessage(__FILE__, __LINE__)
#define LOG_WARNING LOG_INFO
#define LOG_ERROR LOG_INFO
#define LOG_FATAL ::dmlc::LogMessageFatal(__FILE__, __LINE__)
#define LOG(severity) LOG_##severity.stream()

#define CHECK(x) \
  if (!(x)) LOG(FATAL) << "Check failed: " #x << ' '
#define DMLC_SHIM_CHECK_OP(op, a, b) \
  if (!((a)op(b))) LOG(FATAL) << "Check failed: " #a " " #op " " #b << " (" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) DMLC_SHIM_CHECK_OP(==, a, b)
#define CHECK_NE(a, b) DMLC_SHIM_CHECK_OP(!=, a, b)
#define CHECK_LT(a, b) DMLC_SHIM_CHECK_OP(<, a, b)
#define CHECK_LE(a, b) DMLC_SHIM_CHECK_OP(<=, a, b)
#define CHECK_GT(a, b) DMLC_SHIM_CHECK_OP(>, a, b)
#define CHECK_GE(a, b) DMLC_SHIM_CHECK_OP(>=, a, b)
#define CHECK_NOTNULL(x) \
  ((x) == nullptr ? (LOG(FATAL) << "Check notnull: " #x << ' ', (x)) : (x))
#define DCHECK(x) CHECK(x)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK_NE(a, b) CHECK_NE(a, b)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
