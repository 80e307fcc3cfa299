// rt_device_internal.h — what the library's other translation units may know about an rt_ctx (defined in rt_device.hip).
#ifndef RT_DEVICE_INTERNAL_H
#define RT_DEVICE_INTERNAL_H
#include "rt_api.h"
namespace rtk_internal {
int ctx_device(const rt_ctx *ctx);                                 // HIP device id of the context
int ctx_fail(rt_ctx *ctx, int status, const char *fmt, ...);       // sets rt_last_error(ctx) (ctx == NULL: the creation error), returns status
}
#endif
