// Issue cost of the vector instructions the tracing kernels are made of, on gfx950, at 1 / 2 / 4 wavefronts per SIMD.
// A measurement tool, not part of the product: `hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/build/valu_rates`.
// Every test is a loop of 64 iterations over 32 instructions written in inline asm (independent destinations, so
// the stream is bound by issue, not by dependencies); cycles come from s_memtime on one wavefront per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

#define KERNEL(name, body, clobbers...)                                                                      \
    __global__ void __launch_bounds__(1024) name(unsigned long long* out, float* sink, int iters) {          \
        extern __shared__ float lds[];                                                                        \
        float a = threadIdx.x * 1.0001f + 1.0f, b = a * 0.5f, c = b + 3.0f;                                   \
        unsigned ua = threadIdx.x * 2654435761u + 12345u, ub = ua ^ 0x9E3779B9u;                              \
        lds[threadIdx.x] = a;                                                                                 \
        __syncthreads();                                                                                      \
        unsigned addr = (threadIdx.x * 4u) & 4095u;                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                           \
        for (int i = 0; i < iters; ++i) {                                                                     \
            asm volatile(REP32(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(ua), "+v"(ub) : "v"(addr) : clobbers); \
        }                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)");                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                           \
        if ((threadIdx.x & 63u) == 0u) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                  \
        if (a == 12345.678f && ua == 77u) sink[0] = a + b + c + (float)ub;                                    \
    }

KERNEL(k_fma, "v_fma_f32 v40, %0, %1, %2\n", "v40")
KERNEL(k_add, "v_add_f32 v40, %0, %1\n", "v40")
KERNEL(k_pk_fma, "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[46:47]\n", "v40", "v41")
KERNEL(k_pk_mul, "v_pk_mul_f32 v[40:41], v[42:43], v[44:45]\n", "v40", "v41")
KERNEL(k_mad64, "v_mad_u64_u32 v[40:41], s[10:11], %3, %4, v[42:43]\n", "v40", "v41", "s10", "s11")
KERNEL(k_mul_lo, "v_mul_lo_u32 v40, %3, %4\n", "v40")
KERNEL(k_mul_hi, "v_mul_hi_u32 v40, %3, %4\n", "v40")
KERNEL(k_mul24, "v_mul_u32_u24 v40, %3, %4\n", "v40")
KERNEL(k_mad24, "v_mad_u32_u24 v40, %3, %4, %3\n", "v40")
KERNEL(k_xor3, "v_xor_b32 v40, %3, %4\n", "v40")
KERNEL(k_add3, "v_add3_u32 v40, %3, %4, %3\n", "v40")
KERNEL(k_and, "v_and_b32 v40, %3, %4\n", "v40")
KERNEL(k_perm, "v_perm_b32 v40, %3, %4, %3\n", "v40")
KERNEL(k_cvt, "v_cvt_f32_u32 v40, %3\n", "v40")
KERNEL(k_cvt_sdwa, "v_cvt_f32_u32_sdwa v40, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n", "v40")
KERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte2 v40, %3\n", "v40")
KERNEL(k_max3, "v_max3_f32 v40, %0, %1, %2\n", "v40")
KERNEL(k_max, "v_max_f32 v40, %0, %1\n", "v40")
KERNEL(k_cndmask, "v_cndmask_b32 v40, %0, %1, vcc\n", "v40")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1\n", "vcc")
KERNEL(k_rcp, "v_rcp_f32 v40, %0\n", "v40")
KERNEL(k_rsq, "v_rsq_f32 v40, %0\n", "v40")
KERNEL(k_sqrt, "v_sqrt_f32 v40, %0\n", "v40")
KERNEL(k_exp, "v_exp_f32 v40, %0\n", "v40")
KERNEL(k_log, "v_log_f32 v40, %0\n", "v40")
KERNEL(k_sin, "v_sin_f32 v40, %0\n", "v40")
KERNEL(k_div_scale, "v_div_scale_f32 v40, vcc, %0, %1, %0\n", "v40", "vcc")
KERNEL(k_div_fmas, "v_div_fmas_f32 v40, %0, %1, %2\n", "v40")
KERNEL(k_div_fixup, "v_div_fixup_f32 v40, %0, %1, %2\n", "v40")
KERNEL(k_fma64, "v_fma_f64 v[40:41], v[42:43], v[44:45], v[46:47]\n", "v40", "v41")
KERNEL(k_readlane, "v_readlane_b32 s10, %0, 3\n", "s10")
KERNEL(k_writelane, "v_writelane_b32 v40, s4, 3\n", "v40")
KERNEL(k_mov_dpp, "v_mov_b32_dpp v40, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n", "v40")
KERNEL(k_ds_read, "ds_read_b32 v40, %5\n", "v40")
KERNEL(k_ds_read128, "ds_read_b128 v[40:43], %5\n", "v40", "v41", "v42", "v43")
KERNEL(k_ds_bperm, "ds_bpermute_b32 v40, %5, %0\n", "v40")
KERNEL(k_salu, "s_add_u32 s10, s10, s11\n", "s10", "scc")
KERNEL(k_mix, "v_fma_f32 v40, %0, %1, %2\ns_add_u32 s10, s10, s11\n", "v40", "s10", "scc")
KERNEL(k_ballot, "v_cmp_lt_f32 vcc, %0, %1\ns_bcnt1_i32_b64 s10, vcc\n", "vcc", "s10", "scc")

struct Test { const char* name; void (*fn)(unsigned long long*, float*, int); int perBody; };

int main() {
    std::vector<Test> tests = {
        {"v_fma_f32", k_fma, 1}, {"v_add_f32", k_add, 1}, {"v_pk_fma_f32", k_pk_fma, 1}, {"v_pk_mul_f32", k_pk_mul, 1},
        {"v_mad_u64_u32", k_mad64, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1}, {"v_mul_u32_u24", k_mul24, 1},
        {"v_mad_u32_u24", k_mad24, 1}, {"v_xor_b32", k_xor3, 1}, {"v_add3_u32", k_add3, 1}, {"v_and_b32", k_and, 1},
        {"v_perm_b32", k_perm, 1}, {"v_cvt_f32_u32", k_cvt, 1}, {"v_cvt_f32_u32_sdwa", k_cvt_sdwa, 1}, {"v_cvt_f32_ubyte2", k_cvt_ubyte, 1},
        {"v_max3_f32", k_max3, 1}, {"v_max_f32", k_max, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp_lt_f32", k_cmp, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_exp_f32", k_exp, 1}, {"v_log_f32", k_log, 1},
        {"v_sin_f32", k_sin, 1}, {"v_div_scale_f32", k_div_scale, 1}, {"v_div_fmas_f32", k_div_fmas, 1}, {"v_div_fixup_f32", k_div_fixup, 1},
        {"v_fma_f64", k_fma64, 1}, {"v_readlane_b32", k_readlane, 1}, {"v_writelane_b32", k_writelane, 1}, {"v_mov_b32_dpp", k_mov_dpp, 1},
        {"ds_read_b32", k_ds_read, 1}, {"ds_read_b128", k_ds_read128, 1}, {"ds_bpermute_b32", k_ds_bperm, 1},
        {"s_add_u32", k_salu, 1}, {"v_fma+s_add pair", k_mix, 1}, {"v_cmp+s_bcnt pair", k_ballot, 1},
    };
    unsigned long long* out;
    float* sink;
    (void)hipMalloc(&out, 256 * 16 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 64);
    const int iters = 64;
    const size_t ldsBytes = 100 * 1024; // one workgroup per CU
    std::printf("cycles (s_memtime ticks) per wave-instruction and SIMD at 1 / 2 / 4 wavefronts per SIMD; [ticks one wave sees per own instruction]\n");
    for (auto& t : tests) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(t.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        std::printf("%-22s", t.name);
        for (int wps : {1, 2, 4}) {
            const int block = 256 * wps;
            std::vector<unsigned long long> h(256 * 16);
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(out, 0, 256 * 16 * sizeof(unsigned long long));
                hipLaunchKernelGGL(t.fn, dim3(256), dim3(block), ldsBytes, 0, out, sink, iters);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                double sum = 0; int n = 0;
                for (int bI = 0; bI < 256; ++bI) for (int w = 0; w < block / 64; ++w) { sum += (double)h[bI * 16 + w]; ++n; }
                const double perWave = sum / n / (iters * 32.0); // ticks a wave sees per own instruction
                if (perWave < best) best = perWave;
            }
            std::printf("  %6.2f [%6.2f]", best / wps, best);
        }
        std::printf("\n");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { std::printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
