// Communicator handle (see comm.cpp).
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

struct polee_comm {
    polee_ctx *ctx = nullptr;
    void *comm = nullptr;  // ncclComm_t (null: host-staged communicator)
    polee_host_allreduce_fn host_allreduce = nullptr;  // polee_comm_create_host
    void *host_user = nullptr;
    std::vector<uint8_t> staging;
    int32_t nranks = 1, rank = 0;
    int refs = 1;  // the creator's reference + one per VI handle using it
};

namespace polee {
// in-place sum over ranks of a device buffer, enqueued on the context's stream
polee_status comm_allreduce_device(polee_comm *c, void *buf, size_t count, bool f64);
}  // namespace polee
