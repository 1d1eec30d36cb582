// internal.h — what the translation units of librfwhip.so share besides the public C ABI (not installed, not exported).
#pragma once

// sets the calling thread's rfwhip_last_error() text and returns `code` (rfwhip_api.cpp)
int rfwhip_internal_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
