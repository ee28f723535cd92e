// Training entry points of libscs (include/scs.h) - placeholder until the training kernels land.
#include "scs_ctx.h"

namespace scs {
void train_free(scs_ctx*) {}
}

extern "C" {
int scs_train_step(scs_handle, const float*, const float*, const float*, const float*, const uint8_t*, int64_t,
                   uint64_t, const uint8_t*, float*, void*) {
    scs::set_error("scs_train_step: not built yet");
    return SCS_ERR_STATE;
}
int scs_adam_apply(scs_handle, float, float, float, float, float, void*) {
    scs::set_error("scs_adam_apply: not built yet");
    return SCS_ERR_STATE;
}
int scs_eval_batch(scs_handle, const float*, const float*, const float*, const float*, const uint8_t*, int64_t,
                   double*, int64_t*, void*) {
    scs::set_error("scs_eval_batch: not built yet");
    return SCS_ERR_STATE;
}
}
