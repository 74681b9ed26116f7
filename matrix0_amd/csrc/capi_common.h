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
