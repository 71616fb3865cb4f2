// crn_internal.h — error plumbing shared by the host translation units of libcrnsense.
#ifndef CRN_INTERNAL_H
#define CRN_INTERNAL_H
#include <string>

namespace crn {
// Records the thread-local message returned by crn_last_error() and returns `code`.
int fail(int code, const std::string &msg);
}  // namespace crn
#endif
