// pi_host.h -- what the library's translation units share on the HOST side (pi_abi.hip defines it, pi_s1_abi.hip uses it):
// the residency guard of the resident ("persistent") launches.  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace pi_host {

struct Resident {
    int dev = 0, slot = 0;
    volatile int* hs = nullptr;       // host-mapped status slot {roll call complete, step, task, aborted}
    unsigned char* scratch = nullptr; // per-device scratch, zeroed on the stream: 256 B of sync words | granule outbox
    unsigned long long timeout_ticks = 0, first_timeout_ticks = 0;   // bounds of a hand-over wait (100 MHz ticks): options persist_*_ms
};

// hipSuccess: go ahead and launch on `stream`; anything else: take the launch-per-step path (resident launches switched off or
// aborted earlier on this device, another stream holds a resident grid, the stream is being captured, no memory)
hipError_t resident_begin(void* stream, size_t outbox_bytes, Resident& r);
// after the launch: waits for the roll call (option persist_handshake).  hipSuccess: resident and running;
// hipErrorLaunchFailure: it aborted -- recompute launch by launch (the device keeps that path until persist_reset)
hipError_t resident_launched(void* stream, Resident& r, unsigned grid, const char* what);
// PERCNN_PI_EASYNC once if a resident launch nobody waited for has aborted since the last look, else 0
int resident_async_error();
int resident_cu_count();

}  // namespace pi_host
