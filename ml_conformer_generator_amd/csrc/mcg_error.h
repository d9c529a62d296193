// Status codes + the thread-local error text of libmlconfgen_hip.so.  No HIP types here: the host-only
// translation units (mcg_plan_host.cpp, built with AddressSanitizer by `make asan`) include just this.
#pragma once

#define MCG_OK 0
#define MCG_ERR_ARG 1
#define MCG_ERR_HIP 2
#define MCG_ERR_STATE 3

extern "C" void mcg_set_error(const char* fmt, ...);
