/*
 * bdpt_ctx_internal.h -- what other translation units of the library may ask a context
 * (bdpt_api.cpp owns the struct).
 */
#ifndef BDPT_CTX_INTERNAL_H
#define BDPT_CTX_INTERNAL_H

#include <hip/hip_runtime.h>

#include "../../include/bdpt.h"

int bdpt_ctx_device(const bdpt_ctx *ctx);
hipStream_t bdpt_ctx_stream(const bdpt_ctx *ctx);

#endif
