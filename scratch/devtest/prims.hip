// device vs host check of bs_core.h primitives (scratch; run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../rust_mdbg_amd/csrc/bs_core.h"
__global__ void k(const unsigned* m, const unsigned* x0, const unsigned* x1, unsigned* o0, unsigned* o1, unsigned* sel, unsigned* orr, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    unsigned a = x0[i], b = x1[i]; bs_compress2(m[i], a, b); o0[i] = a; o1[i] = b;
    unsigned pc = bs_popc(m[i]); sel[i] = pc ? bs_select_msb(m[i], x0[i] % pc) : 99;
    orr[i] = bs_or3i(m[i], x0[i], x1[i], i & 7);
}
int main() {
    const int n = 1 << 20;
    std::vector<unsigned> m(n), x0(n), x1(n), o0(n), o1(n), sel(n), orr(n);
    srand(5);
    for (int i = 0; i < n; ++i) { m[i] = (unsigned)rand() * 2654435761u ^ (unsigned)rand(); if (i % 17 == 0) m[i] = 0xFFFFFFFFu; if (i % 19 == 0) m[i] = 0; if (i % 23 == 0) m[i] |= 0xFFFF0000u; x0[i] = (unsigned)rand() * 40503u ^ (unsigned)rand() << 3; x1[i] = (unsigned)rand() * 2246822519u ^ (unsigned)rand(); }
    unsigned *dm, *d0, *d1, *p0, *p1, *ps, *po;
    hipMalloc(&dm, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&p0, n * 4); hipMalloc(&p1, n * 4); hipMalloc(&ps, n * 4); hipMalloc(&po, n * 4);
    hipMemcpy(dm, m.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d0, x0.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d1, x1.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dm, d0, d1, p0, p1, ps, po, n);
    hipMemcpy(o0.data(), p0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), p1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(sel.data(), ps, n * 4, hipMemcpyDeviceToHost); hipMemcpy(orr.data(), po, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        unsigned a = x0[i], b = x1[i]; bs_compress2(m[i], a, b);
        unsigned pc = bs_popc(m[i]); unsigned s = pc ? bs_select_msb(m[i], x0[i] % pc) : 99;
        unsigned o = bs_or3i(m[i], x0[i], x1[i], i & 7);
        if (a != o0[i] || b != o1[i] || s != sel[i] || o != orr[i]) { if (bad < 10) printf("i %d m %08x x0 %08x x1 %08x: dev %08x %08x sel %u or %08x | host %08x %08x sel %u or %08x\n", i, m[i], x0[i], x1[i], o0[i], o1[i], sel[i], orr[i], a, b, s, o); ++bad; }
    }
    printf("prims: %d mismatches of %d\n", bad, n);
    return bad != 0;
}
