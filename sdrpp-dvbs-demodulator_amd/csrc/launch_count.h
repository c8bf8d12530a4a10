// Every kernel launch of the library goes through hipLaunchKernelGGL: this header counts them (a process-wide relaxed counter, read through
// dvbs2gpu_get_state(ctx, "kernel_launches")) so that bench.py can say how many launches one drop-in call costs.  Host-side only.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
namespace s2 { extern std::atomic<long long> g_kernel_launches; }
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...) do { s2::g_kernel_launches.fetch_add(1, std::memory_order_relaxed); hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__); } while (0)
