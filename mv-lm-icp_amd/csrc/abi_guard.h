// No C++ exception may cross the C ABI (include/mvicp.h): every exported function with a body that can allocate is a
// function-try-block ending in MVICP_GUARD_ABI, which turns whatever was thrown (std::bad_alloc from a host vector, std::system_error
// from a host thread, ...) into MVICP_ERR_INTERNAL + mvicp_last_error().  Host-only header (no HIP types).
#pragma once

namespace mvicp {
int abi_exception() noexcept;   // api.cpp: call ONLY from inside a catch block (it rethrows to classify)
}

#define MVICP_GUARD_ABI catch (...) { return mvicp::abi_exception(); }
