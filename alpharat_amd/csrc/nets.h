// Device policy/value heads -- placeholder until the HIP evaluators land (next commit): the entry
// points exist and fail loudly, they never fall back to another evaluator.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/alpharat_hip.h"
#include "dev_search.h"

struct ArNet {
    int device;
};

static int nets_fail(int code, const std::string& msg);

template <int NW>
static int net_forward_queue(ArNet*, const ar::LeafReq<NW>*, const uint32_t*, uint32_t, const ar::Slot<NW>*,
                             const uint8_t*, ar::EvalOut*, hipStream_t) {
    return nets_fail(AR_E_BACKEND, "network evaluators are not built in this revision");
}
