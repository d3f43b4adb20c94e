// development aid: does RCCL initialise on this box? (hipcc tools/rccl_probe.cpp -lrccl -o /tmp/rccl_probe)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
int main()
{
    int n = 0;
    printf("hipGetDeviceCount -> %d, n=%d\n", (int)hipGetDeviceCount(&n), n);
    printf("hipSetDevice -> %d\n", (int)hipSetDevice(0));
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    printf("ncclGetUniqueId -> %d (%s)\n", (int)r, ncclGetErrorString(r));
    ncclComm_t comm;
    r = ncclCommInitRank(&comm, 1, id, 0);
    printf("ncclCommInitRank -> %d (%s)\n", (int)r, ncclGetErrorString(r));
    if (r == ncclSuccess) {
        long *d; hipMalloc((void**)&d, 64);
        hipMemset(d, 0, 64);
        r = ncclAllReduce(d, d + 4, 4, ncclInt64, ncclSum, comm, 0);
        printf("ncclAllReduce -> %d, sync %d\n", (int)r, (int)hipDeviceSynchronize());
        ncclCommDestroy(comm);
    }
    return 0;
}
