// spin.hip -- a kernel that lasts a given time and stamps its start and end (100 MHz wall clock):
// the probe tools/graph_order.py uses to learn how the hipGraph executor orders the two branches
// of a captured two-stream graph.  hipcc --offload-arch=gfx950 -shared -fPIC -o libspin.so spin.hip
#include <hip/hip_runtime.h>

__global__ void spin_kernel(long long ticks, long long *stamps, int slot)
{
    const long long t0 = wall_clock64();
    long long t = t0;
    // bounded: at most ~50 ms whatever is asked
    const long long lim = ticks < 5000000 ? ticks : 5000000;
    while (t - t0 < lim) t = wall_clock64();
    if (threadIdx.x == 0) {
        stamps[2 * slot] = t0;
        stamps[2 * slot + 1] = t;
    }
}

extern "C" int spin_launch(double us, long long *stamps, int slot, void *stream)
{
    spin_kernel<<<1, 64, 0, (hipStream_t)stream>>>((long long)(us * 100.0), stamps, slot);
    return (int)hipGetLastError();
}
