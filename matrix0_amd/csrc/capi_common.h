// Shared between the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/m0_engine.h"

class Net;
void m0_set_error(const std::string& s);
Net* m0_net_impl(m0_net* n);
hipStream_t m0_net_stream(m0_net* n);
int m0_net_device(m0_net* n);
// The handle's mutex: every forward on a network (m0_net_infer from any thread, an engine stepping on it) runs under it --
// the workspace and the stream belong to the handle.
void m0_net_lock(m0_net* n);
void m0_net_unlock(m0_net* n);
