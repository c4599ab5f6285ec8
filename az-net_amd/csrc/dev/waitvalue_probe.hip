// dev probe: hipStreamWaitValue32 / hipStreamWriteValue32 across two streams, enqueue order reversed (wait first)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_mark(int *p, int v) { if (threadIdx.x == 0) atomicExch(p, v); }
int main()
{
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned *flag; int *order;
    hipError_t e = hipExtMallocWithFlags((void **)&flag, 64, hipMallocSignalMemory);
    printf("signal memory: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { hipMalloc((void **)&flag, 64); }
    hipMalloc((void **)&order, 64);
    hipMemset(flag, 0, 64); hipMemset(order, 0, 64);
    hipDeviceSynchronize();
    // stream a: wait(flag >= 7) then mark order[1] = 2; stream b (enqueued later): mark order[0] = 1, write flag = 7
    e = hipStreamWaitValue32(a, flag, 7, hipStreamWaitValueGte, 0xFFFFFFFFu);
    printf("wait enqueue: %s\n", hipGetErrorString(e));
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, a, order + 1, 2);
    int h[2] = {-1, -1};
    hipMemcpy(h, order, 8, hipMemcpyDeviceToHost);
    printf("before the write: order = %d %d (expect 0 0)\n", h[0], h[1]);
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, b, order, 1);
    e = hipStreamWriteValue32(b, flag, 7, 0);
    printf("write enqueue: %s\n", hipGetErrorString(e));
    hipStreamSynchronize(b); hipStreamSynchronize(a);
    hipMemcpy(h, order, 8, hipMemcpyDeviceToHost);
    printf("after: order = %d %d (expect 1 2)\n", h[0], h[1]);
    // latency of a write -> wait hand-over
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, a);
        for (int i = 0; i < 100; ++i) {
            hipStreamWriteValue32(a, flag, 100 + 2 * i + 1000 * it, 0);
            hipStreamWaitValue32(b, flag, 100 + 2 * i + 1000 * it, hipStreamWaitValueGte, 0xFFFFFFFFu);
            hipStreamWriteValue32(b, flag, 101 + 2 * i + 1000 * it, 0);
            hipStreamWaitValue32(a, flag, 101 + 2 * i + 1000 * it, hipStreamWaitValueGte, 0xFFFFFFFFu);
        }
        hipEventRecord(e1, a); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("200 hand-overs: %.3f ms (%.2f us each)\n", ms, ms * 5);
    }
    return 0;
}
