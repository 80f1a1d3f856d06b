// Cost of the reduction tail: B co-resident workgroups each end with one (or two) atomics.  Same address vs slotted addresses, f64 vs u32,
// with / without a returned value.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tail_atomics.hip -o tail_atomics.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(double *acc, unsigned *tick, double *sink, int slots)
{
    double v = threadIdx.x * 1e-9 + blockIdx.x;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (threadIdx.x == 0) {
        if (MODE == 0) unsafeAtomicAdd(acc, v);                                   // one address
        if (MODE == 1) { unsafeAtomicAdd(acc, v); unsafeAtomicAdd(acc + 1, v); }  // two scalars, neighbours (the CG update)
        if (MODE == 2) unsafeAtomicAdd(acc + 16 * (blockIdx.x % slots), v);       // slotted, 128 B apart
        if (MODE == 3) { unsigned t = atomicAdd(tick, 1u); if (t == gridDim.x - 1) sink[0] = v; }      // ticket with return
        if (MODE == 4) atomicAdd(tick, 1u);                                       // integer, no return
        if (MODE == 5) { unsafeAtomicAdd(acc + 16 * (blockIdx.x % slots), v); __threadfence(); unsigned t = atomicAdd(tick, 1u);
                         if (t == gridDim.x - 1) { double s = 0; for (int i = 0; i < slots; ++i) s += acc[16 * i]; sink[0] = s; *tick = 0; } } // slots + last block folds
        if (MODE == 6) sink[blockIdx.x] = v;                                      // plain store of the partial
    }
}

int main()
{
    double *acc, *sink; unsigned *tick;
    CK(hipMalloc(&acc, 1 << 16)); CK(hipMalloc(&sink, 1 << 20)); CK(hipMalloc(&tick, 64));
    CK(hipMemset(acc, 0, 1 << 16)); CK(hipMemset(tick, 0, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[7] = {"f64 one address", "f64 two addresses (same line)", "f64 slotted", "u32 ticket (returned)", "u32 no return", "slots + ticket + fold", "plain store"};
    for (int B : {256, 1024, 2048, 8192}) {
        for (int mode = 0; mode < 7; ++mode)
            for (int slots : {8, 64}) {
                if (slots == 64 && mode != 2 && mode != 5) continue;
                auto go = [&]() {
                    switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(B), dim3(256), 0, 0, acc, tick, sink, slots); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(B), dim3(256), 0, 0, acc, tick, sink, slots); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(B), dim3(256), 0, 0, acc, tick, sink, slots); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(B), dim3(256), 0, 0, acc, tick, sink, slots); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(B), dim3(256), 0, 0, acc, tick, sink, slots); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(B), dim3(256), 0, 0, acc, tick, sink, slots); break;
                    default: hipLaunchKernelGGL(k<6>, dim3(B), dim3(256), 0, 0, acc, tick, sink, slots); break;
                    }
                };
                for (int i = 0; i < 5; ++i) go();
                CK(hipMemset(tick, 0, 64));
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 50; ++i) go();
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("blocks %5d  %-32s slots %2d  %.2f us per launch\n", B, names[mode], slots, ms * 1000 / 50);
                CK(hipMemset(tick, 0, 64));
            }
    }
    return 0;
}
