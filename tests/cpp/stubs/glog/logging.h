// Logging stub for tools/check_reference_compiles.sh ONLY (a syntax / link check of the drop-in claim, container-only):
// glog's macros as stream-swallowing expressions.  Nothing built with it is ever run or used as an oracle.
#pragma once
#include <iostream>
#include <sstream>
#include <string>
namespace statmc_glog_stub {
struct Sink {
    template <class T> Sink &operator<<(const T &) { return *this; }
    Sink &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
struct Voidify { void operator&(Sink &) {} };
}
#define STATMC_GLOG_SINK() statmc_glog_stub::Sink()
#define LOG(x) STATMC_GLOG_SINK()
#define VLOG(x) STATMC_GLOG_SINK()
#define LOG_IF(x, c) STATMC_GLOG_SINK()
#define CHECK(c) (void)(c), STATMC_GLOG_SINK()
#define CHECK_OP_(a, b) (void)(a), (void)(b), STATMC_GLOG_SINK()
#define CHECK_EQ(a, b) CHECK_OP_(a, b)
#define CHECK_NE(a, b) CHECK_OP_(a, b)
#define CHECK_LT(a, b) CHECK_OP_(a, b)
#define CHECK_LE(a, b) CHECK_OP_(a, b)
#define CHECK_GT(a, b) CHECK_OP_(a, b)
#define CHECK_GE(a, b) CHECK_OP_(a, b)
#define CHECK_NOTNULL(p) (p)
#define DCHECK(c) CHECK(c)
#define DCHECK_EQ(a, b) CHECK_OP_(a, b)
#define DCHECK_NE(a, b) CHECK_OP_(a, b)
#define DCHECK_LT(a, b) CHECK_OP_(a, b)
#define DCHECK_LE(a, b) CHECK_OP_(a, b)
#define DCHECK_GT(a, b) CHECK_OP_(a, b)
#define DCHECK_GE(a, b) CHECK_OP_(a, b)
namespace google { inline void InitGoogleLogging(const char *) {} }
