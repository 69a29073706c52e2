// cwr_engine.hip -- host side of the transport engine: device memory, per-step orchestration, the
// batched BiCGSTAB driver, the optional RCCL halo exchange, and the C ABI of include/cwr_transport.h.
// Built for gfx950 only:  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -shared -fPIC
#include "../../include/cwr_transport.h"
#include "cwr_kernels.hpp"
#include "cwr_host_builders.hpp"

#include <dlfcn.h>
#include <atomic>
#include <map>
#include <thread>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace cwr;


// One translation unit, cut along its section banners (round 6): every part below continues the anonymous namespace / the extern "C" block of the
// part before it -- include order is definition order.  Build: build.py compiles THIS file; the parts are its dependencies.
#include "cwr_engine_state.hpp"
#include "cwr_engine_launch.hpp"
#include "cwr_engine_flow.hpp"
#include "cwr_engine_tiling.hpp"
#include "cwr_engine_solve.hpp"
#include "cwr_engine_abi_create.hpp"
#include "cwr_engine_abi_window.hpp"
#include "cwr_engine_abi_step.hpp"
#include "cwr_engine_abi_output.hpp"
#include "cwr_engine_abi_comm.hpp"
