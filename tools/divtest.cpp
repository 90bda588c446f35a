// divtest.cpp -- checks runtime 64-bit division/modulo in device code against the host
// (written while chasing a wrong-result bug in an earlier tensor kernel).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint64_t u64;
__global__ void divk(u64 *q, u64 *r, u64 pn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    q[gid] = gid / pn;
    r[gid] = gid % pn;
}
struct DevMod { u64 p, p2, mu, bh, bl; uint32_t k, pad; };
// the old tensor-kernel addressing, reduced
__global__ void oldaddr(const u64 *__restrict__ ext, u64 *__restrict__ t, const DevMod *__restrict__ mods,
                        uint32_t nmod, uint32_t logn, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const u64 pn = (u64)nmod << logn;
    const u64 b = gid / pn, off = gid % pn;
    const DevMod m = mods[off >> logn];
    const u64 *e = ext + b * 4 * pn + off;
    const u64 c00 = e[0], c01 = e[pn], c10 = e[2 * pn], c11 = e[3 * pn];
    u64 *o = t + b * 3 * pn + off;
    o[0] = c00 + c10 + m.p;
    o[pn] = c00 + c11 + c01 + c10;
    o[2 * pn] = c01 + c11;
}
int main() {
    const u64 pn = 73728, total = 3 * pn;
    u64 *q, *r;
    hipMalloc(&q, total * 8); hipMalloc(&r, total * 8);
    hipLaunchKernelGGL(divk, dim3((total + 255) / 256), dim3(256), 0, 0, q, r, pn, total);
    std::vector<u64> hq(total), hr(total);
    hipMemcpy(hq.data(), q, total * 8, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), r, total * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (u64 i = 0; i < total; i++) if (hq[i] != i / pn || hr[i] != i % pn) bad++;
    printf("div/mod mismatches: %zu of %llu\n", bad, (unsigned long long)total);
    // old addressing
    const uint32_t nmod = 9, logn = 13; const u64 nb = 2, tot2 = nb * pn;
    std::vector<u64> hext(nb * 4 * pn), hmods(nmod * 6);
    for (size_t i = 0; i < hext.size(); i++) hext[i] = i * 2654435761ull;
    std::vector<DevMod> hm(nmod); for (uint32_t i = 0; i < nmod; i++) hm[i] = DevMod{1000 + i, 0, 0, 0, 0, 0, 0};
    u64 *dext, *dt; DevMod *dm;
    hipMalloc(&dext, hext.size() * 8); hipMalloc(&dt, nb * 3 * pn * 8); hipMalloc(&dm, sizeof(DevMod) * nmod);
    hipMemcpy(dext, hext.data(), hext.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dm, hm.data(), sizeof(DevMod) * nmod, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(dt, 0, nb * 3 * pn * 8);
        hipLaunchKernelGGL(oldaddr, dim3((tot2 + 255) / 256), dim3(256), 0, 0, dext, dt, dm, nmod, logn, tot2);
        std::vector<u64> ht(nb * 3 * pn);
        hipMemcpy(ht.data(), dt, ht.size() * 8, hipMemcpyDeviceToHost);
        size_t bad2 = 0;
        for (u64 b = 0; b < nb; b++) for (u64 off = 0; off < pn; off++) {
            const u64 *e = &hext[b * 4 * pn + off];
            u64 w0 = e[0] + e[2 * pn] + hm[off >> logn].p, w1 = e[0] + e[3 * pn] + e[pn] + e[2 * pn], w2 = e[pn] + e[3 * pn];
            const u64 *o = &ht[b * 3 * pn + off];
            if (o[0] != w0 || o[pn] != w1 || o[2 * pn] != w2) bad2++;
        }
        printf("old tensor addressing mismatches (rep %d): %zu\n", rep, bad2);
    }
    return 0;
}
