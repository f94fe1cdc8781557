// dev probe (round 6): how fast can a CU pull L2-resident / MALL-resident / HBM lines into LDS?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/staging_probe.hip -o tools/probes/libstaging_probe.so
// Each block (256 or 512 threads) copies `slab` bytes from src + (block * stride) % span into its LDS, `reps` times, by
//   mode 0: global_load_dwordx4 -> registers -> ds_write_b128 (8 loads in flight per thread)
//   mode 1: buffer_load_dwordx4 ... lds (LDS-DMA), 8 instructions in flight per wave
// and writes one word so that nothing is optimised away.  Host times the launch with events.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

extern "C" __global__ __launch_bounds__(512) void k_probe(const float* __restrict__ src, float* __restrict__ out, size_t span_bytes,
                                                           size_t stride_bytes, int slab_bytes, int reps, int mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const size_t base = ((size_t)blockIdx.x * stride_bytes) % span_bytes;
    const char* p = (const char*)src + base;
    const int nthr = blockDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
    float acc = 0.f;
    if (mode == 0) {
        const int npieces = slab_bytes / 16;
        for (int r = 0; r < reps; ++r) {
            for (int i0 = 0; i0 < npieces; i0 += nthr * 8) {
                f4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * nthr + tid;
                    v[u] = i < npieces ? *(const f4*)(p + (size_t)i * 16) : f4{0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * nthr + tid;
                    if (i < npieces) *(f4*)(smem + (size_t)i * 4) = v[u];
                }
            }
            __syncthreads();
            acc += smem[(tid * 4 + r) % (slab_bytes / 4)];
            __syncthreads();
        }
    } else {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, slab_bytes, 0x00020000);
        const int ninstr = slab_bytes / 1024;
        for (int r = 0; r < reps; ++r) {
            for (int j = wv; j < ninstr; j += nw)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((char*)smem + j * 1024), 16, (unsigned)(j * 1024 + lane * 16), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc += smem[(tid * 4 + r) % (slab_bytes / 4)];
            __syncthreads();
        }
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

extern "C" int probe_launch(const void* src, void* out, size_t span, size_t stride, int slab, int reps, int mode, int blocks, int threads,
                            void* stream) {
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), slab, (hipStream_t)stream, (const float*)src, (float*)out, span, stride, slab, reps, mode);
    return (int)hipGetLastError();
}
