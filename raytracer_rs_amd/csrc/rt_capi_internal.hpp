// Shared by the translation units that implement the C ABI.
#pragma once

#include <string>

namespace rt {

// Records `msg` as the calling thread's last error (rt_last_error) and returns `code`.
int capi_set_error(int code, const std::string &msg);

}  // namespace rt
