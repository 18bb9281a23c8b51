// Latencies the BVH walk lives on, measured on gfx950: a dependent chain of 16-B gathers (one per lane, every lane its own
// chain) through working sets that sit in L1, L2 and beyond, with 1 / 2 / 4 wavefronts per SIMD running the same thing;
// an LDS chain; a VALU dependent chain; the v_cmp -> s_and_saveexec hand-over.
// A measurement tool, not part of the product: `hipcc --offload-arch=gfx950 -O2 tools/mem_latency.hip -o tools/build/mem_latency`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// buf[i].x = index of the next record of the chain record i belongs to
__global__ void __launch_bounds__(1024) chase(const u4* __restrict__ buf, unsigned n, int steps, int sameLine, unsigned long long* out, unsigned* sink) {
    unsigned idx = sameLine ? (blockIdx.x * 131u) % n : (blockIdx.x * 1024u + threadIdx.x) * 97u % n;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < steps; ++i) {
        const u4 v = buf[idx];
        idx = v.x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0u) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (idx == 0xFFFFFFFFu) sink[0] = idx;
}

__global__ void __launch_bounds__(1024) chaseLds(int steps, unsigned long long* out, unsigned* sink) {
    __shared__ unsigned lds[8192];
    for (unsigned i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (i * 1103515245u + 12345u) & 8191u;
    __syncthreads();
    unsigned idx = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < steps; ++i) idx = lds[idx];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0u) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (idx == 0xFFFFFFFFu) sink[0] = idx;
}

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))
#define CHAIN(name, body, clobbers...)                                                                       \
    __global__ void __launch_bounds__(1024) name(int iters, unsigned long long* out, float* sink) {          \
        float a = threadIdx.x * 1.0001f + 1.0f, b = 0.999f, c = 0.001f;                                       \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
        for (int i = 0; i < iters; ++i) asm volatile(REP32(body) : "+v"(a) : "v"(b), "v"(c) : clobbers);      \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
        if ((threadIdx.x & 63u) == 0u) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                  \
        if (a == 12345.678f) sink[0] = a;                                                                     \
    }
CHAIN(c_fma, "v_fma_f32 %0, %0, %1, %2\n", "memory")
CHAIN(c_pkfma, "v_pk_fma_f32 v[40:41], v[40:41], v[42:43], v[44:45]\n", "v40", "v41")
CHAIN(c_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %2, vcc\n", "vcc")
CHAIN(c_cnd_only, "v_cndmask_b32 v40, %0, %1, vcc\n", "v40")
CHAIN(c_cnd_sgpr, "v_cndmask_b32_e64 v40, %0, %1, s[10:11]\n", "v40")
CHAIN(c_cmp_saveexec, "v_cmp_lt_f32 vcc, %1, %0\ns_and_saveexec_b64 s[10:11], vcc\ns_or_b64 exec, exec, s[10:11]\n", "vcc", "s10", "s11", "scc")
CHAIN(c_cmp_bcnt_branch, "v_cmp_lt_f32 vcc, %1, %0\ns_bcnt1_i32_b64 s10, vcc\ns_cmp_gt_u32 s10, 3\ns_cselect_b64 s[12:13], -1, 0\n", "vcc", "s10", "s12", "s13", "scc")
CHAIN(c_sdwa, "v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n", "memory")
CHAIN(c_rcp, "v_rcp_f32 %0, %0\n", "memory")
CHAIN(c_readlane, "v_readlane_b32 s10, %0, 3\nv_add_f32 %0, s10, %0\n", "s10")

struct Res { double one, two, four; };

template <class F>
static double run(F launch, unsigned long long* out, int block, double perUnit) {
    std::vector<unsigned long long> h(256 * 16);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(out, 0, h.size() * 8);
        launch(block);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < block / 64; ++w) { sum += (double)h[b * 16 + w]; ++n; }
        best = std::min(best, sum / n / perUnit);
    }
    return best;
}

int main() {
    unsigned long long* out; unsigned* sink; float* fsink;
    (void)hipMalloc(&out, 256 * 16 * 8); (void)hipMalloc(&sink, 64); (void)hipMalloc(&fsink, 64);
    std::printf("ticks of s_memtime per dependent step, one wave's view, at 1 / 2 / 4 wavefronts per SIMD (256 workgroups)\n");
    const int steps = 2000;
    std::mt19937 rng(12345);
    for (size_t bytes : {size_t(8) << 10, size_t(16) << 10, size_t(24) << 10, size_t(64) << 10, size_t(256) << 10, size_t(2) << 20, size_t(16) << 20, size_t(512) << 20}) {
        const unsigned n = (unsigned)(bytes / 16);
        std::vector<u4> h(n);
        std::vector<unsigned> perm(n);
        for (unsigned i = 0; i < n; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        for (unsigned i = 0; i < n; ++i) h[perm[i]] = u4{perm[(i + 1) % n], 0u, 0u, 0u}; // one cycle through all records
        u4* buf; (void)hipMalloc(&buf, bytes);
        (void)hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice);
        for (int same : {0, 1}) {
            std::printf("16-B gather chain, working set %8zu KB, %s:", bytes >> 10, same ? "all lanes one address " : "every lane its own line");
            for (int wps : {1, 2, 4}) {
                const double t = run([&](int block) { hipLaunchKernelGGL(chase, dim3(256), dim3(block), 0, 0, buf, n, steps, same, out, sink); }, out, 256 * wps, steps);
                std::printf("  %7.1f", t);
            }
            std::printf("\n");
        }
        (void)hipFree(buf);
    }
    std::printf("LDS 4-B chain:");
    for (int wps : {1, 2, 4}) std::printf("  %7.1f", run([&](int block) { hipLaunchKernelGGL(chaseLds, dim3(256), dim3(block), 0, 0, steps, out, sink); }, out, 256 * wps, steps));
    std::printf("\n");
    struct T { const char* name; void (*fn)(int, unsigned long long*, float*); double per; };
    const T tests[] = {{"v_fma_f32 dependent", c_fma, 1}, {"v_pk_fma_f32 dependent", c_pkfma, 1}, {"v_cmp -> v_cndmask dependent pair", c_cmp_cnd, 1},
                       {"v_cndmask (vcc, independent)", c_cnd_only, 1}, {"v_cndmask_e64 (sgpr pair, independent)", c_cnd_sgpr, 1},
                       {"v_cmp -> s_and_saveexec -> s_or exec", c_cmp_saveexec, 1}, {"v_cmp -> s_bcnt1 -> s_cmp -> s_cselect", c_cmp_bcnt_branch, 1},
                       {"v_cvt_f32_u32_sdwa dependent", c_sdwa, 1}, {"v_rcp_f32 dependent", c_rcp, 1}, {"v_readlane -> v_add dependent", c_readlane, 1}};
    for (const T& t : tests) {
        std::printf("%-42s", t.name);
        for (int wps : {1, 2, 4}) std::printf("  %7.2f", run([&](int block) { hipLaunchKernelGGL(t.fn, dim3(256), dim3(block), 0, 0, 64, out, fsink); }, out, 256 * wps, 64 * 32.0));
        std::printf("\n");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { std::printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
