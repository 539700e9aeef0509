// Internal to libmeshdqn_hip.so (not part of the C ABI): what the translation units share besides the public header.
// The library is built from several translation units compiled in parallel (meshdqn_amd/build.py) with
// -fvisibility=hidden: only the MDQ_API entry points of include/meshdqn_hip.h are exported.
#pragma once

// records the text mdq_last_error() returns (thread-local, mdq_ipcs.hip) and returns -2
int mdq_set_error(const char* msg);
