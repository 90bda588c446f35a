// Probe (GPU box): which stream-ordered-allocator / stream-query calls does this HIP runtime survive?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define P(...) do { printf(__VA_ARGS__); fflush(stdout); } while (0)
int main() {
    hipStream_t s;
    P("hipStreamCreate %d\n", (int)hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipMemPoolProps props;
    memset(&props, 0, sizeof(props));
    props.allocType = hipMemAllocationTypePinned;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = 0;
    hipMemPool_t mp = nullptr;
    P("hipMemPoolCreate ...\n");
    hipError_t e = hipMemPoolCreate(&mp, &props);
    P("hipMemPoolCreate -> %d (%s) pool %p\n", (int)e, hipGetErrorString(e), (void *)mp);
    uint64_t keep = ~0ull;
    e = hipMemPoolSetAttribute(mp, hipMemPoolAttrReleaseThreshold, &keep);
    P("hipMemPoolSetAttribute -> %d\n", (int)e);
    void *p = nullptr;
    P("hipMallocFromPoolAsync(stream) ...\n");
    e = hipMallocFromPoolAsync(&p, 1 << 20, mp, s);
    P(" -> %d %p\n", (int)e, p);
    e = hipMemsetAsync(p, 1, 1 << 20, s);
    P("memset -> %d\n", (int)e);
    e = hipFreeAsync(p, s);
    P("hipFreeAsync -> %d\n", (int)e);
    P("hipMallocFromPoolAsync(null stream) ...\n");
    e = hipMallocFromPoolAsync(&p, 1 << 20, mp, nullptr);
    P(" -> %d %p\n", (int)e, p);
    e = hipFreeAsync(p, nullptr);
    P("hipFreeAsync(null) -> %d\n", (int)e);
    e = hipMallocFromPoolAsync(&p, (size_t)3 << 30, mp, s);
    P("3 GiB from pool -> %d %p\n", (int)e, p);
    e = hipFree(p);
    P("hipFree of a pool block -> %d\n", (int)e);
    P("hipStreamQuery(live) -> %d\n", (int)hipStreamQuery(s));
    P("hipStreamSynchronize -> %d\n", (int)hipStreamSynchronize(s));
    P("hipStreamDestroy -> %d\n", (int)hipStreamDestroy(s));
    P("hipStreamQuery(destroyed) ...\n");
    e = hipStreamQuery(s);
    P(" -> %d (%s)\n", (int)e, hipGetErrorString(e));
    P("hipGetLastError -> %d\n", (int)hipGetLastError());
    hipStream_t s2;
    P("hipStreamCreate again %d  same handle: %d\n", (int)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking), (int)(s2 == s));
    e = hipMemPoolTrimTo(mp, 0);
    P("trim -> %d\n", (int)e);
    e = hipMemPoolDestroy(mp);
    P("destroy pool -> %d\n", (int)e);
    return 0;
}
