// la_stubs.cpp -- entry points declared in include/lyricalign.h whose kernels are not written yet.
// Each returns LA_EUNSUPPORTED; entries are deleted from here as the real kernels land.
#include "la_common.h"

#define LA_STUB(name) la::set_error(#name ": not implemented yet"); return LA_EUNSUPPORTED

extern "C" int la_logmel_workspace_bytes(int32_t, int32_t, size_t *) { LA_STUB(la_logmel_workspace_bytes); }
extern "C" int la_logmel_f32(const float *, int32_t, int32_t, const float *, const float *, float *, int64_t, int64_t,
                             void *, size_t, void *) { LA_STUB(la_logmel_f32); }
