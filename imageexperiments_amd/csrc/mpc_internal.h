// mpc_internal.h -- product: what the translation units of libmpcodec.so share besides the public C ABI (include/mpcodec.h).
#pragma once

// the text mpc_last_error() returns on the calling thread (mpcodec_capi.cpp)
extern "C" void mpc_set_error_text(const char* text);
