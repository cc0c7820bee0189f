// microbench.hip — instruction-rate probes for gfx950 (which integer/FP op should carry the
// 256-bit modular multiply?).  Build: hipcc --offload-arch=gfx950 -O3 bench/microbench.hip -o bench/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../hodor_amd/csrc/fr9.cuh"
#include "../hodor_amd/csrc/fr9w3.cuh"
#include "../hodor_amd/csrc/blake2s.cuh"

using namespace hodor;
#define ITERS 2048
#define ILP 8

__global__ void k_mad64(uint64_t *out, uint32_t a, uint32_t b)
{
    uint64_t acc[ILP];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint64_t r;
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(acc[i]) : "vcc");
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mullo(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mulhi(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad24(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(acc[i]), "v"(y), "v"(acc[i]));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_addc(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(r) : "v"(acc[i]), "v"(y) : "vcc");
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add32(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_lshladd64(uint64_t *out, uint32_t a, uint32_t b)
{
    uint64_t acc[ILP];
    uint64_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint64_t r;
            asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma64(uint64_t *out, uint32_t a, uint32_t b)
{
    double acc[ILP];
    double y = 1.0 + 1e-9 * (b + blockIdx.x), z = 1e-3 * a;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            double r;
            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(acc[i]), "v"(y), "v"(z));
            acc[i] = r;
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

__global__ void k_fma32(uint64_t *out, uint32_t a, uint32_t b)
{
    float acc[ILP];
    float y = 1.0f + 1e-6f * (b + blockIdx.x), z = 1e-3f * a;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            float r;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(acc[i]), "v"(y), "v"(z));
            acc[i] = r;
        }
    }
    float s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

__global__ void k_alignbit(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_alignbit_b32 %0, %1, %1, 12" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_xor(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add3(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_add3_u32 %0, %1, %2, %1" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_perm(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_perm_b32 %0, %1, %1, %2" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_lshrrev64x(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the carry extraction of the 9 x 29 products: acc >>= 29 on a 64-bit accumulator
__global__ void k_lshr64(uint64_t *out, uint32_t a, uint32_t b)
{
    uint64_t acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = ((uint64_t)(a + i + threadIdx.x) << 33) | (b + blockIdx.x);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint64_t r;
            asm volatile("v_lshrrev_b64 %0, 1, %1" : "=v"(r) : "v"(acc[i]));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_alignbit2(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


__global__ void k_xor_sdwa(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_rot16_sdwa(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_rot16_plain(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_xor_b32 %0, %1, %2\n v_alignbit_b32 %0, %0, %0, 16" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// the same xor + rotate pairs, but the ILP xors first and the ILP rotates after them (dependency distance
// ILP instead of 1): does a dependent instruction directly behind its producer cost an issue slot?
__global__ void k_rot16_grouped(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_alignbit_b32 %0, %0, %0, 16" : "+v"(acc[i]));
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// distance 2: xor_i, xor_i+1, align_i, align_i+1
__global__ void k_rot16_dist2(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i += 2) {
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc[i + 1]) : "v"(y));
            asm volatile("v_alignbit_b32 %0, %0, %0, 16" : "+v"(acc[i]));
            asm volatile("v_alignbit_b32 %0, %0, %0, 16" : "+v"(acc[i + 1]));
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_xad(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_xad_u32 %0, %1, %2, %2" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_xor_dpp(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_xor_b32_dpp %0, %1, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add_dpp(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


__global__ void k_pk_swap(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add_e64(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_add_u32_e64 %0, %1, %2" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_xor_e64(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_xor_b32_e64 %0, %1, %2" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pk_add(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_pk_add_u16 %0, %1, %2" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_lshl_or(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_bfi(uint64_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[ILP];
    uint32_t y = b + blockIdx.x;
    for (int i = 0; i < ILP; i++) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            uint32_t r;
            asm volatile("v_bfi_b32 %0, %1, %2, %1" : "=&v"(r) : "v"(acc[i]), "v"(y));
            acc[i] = r;
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define MUL_ITERS 256
__global__ void k_frmul(uint64_t *out, FrParams P, uint32_t seed)
{
    Fr x[2], w;
    for (int i = 0; i < 8; i++) { x[0].v[i] = seed + i + threadIdx.x; x[1].v[i] = seed * 3 + i + blockIdx.x; w.v[i] = seed * 7 + i; }
    x[0].v[7] &= 0x0fffffff; x[1].v[7] &= 0x0fffffff; w.v[7] &= 0x0fffffff;
    for (int it = 0; it < MUL_ITERS; it++) {
        x[0] = fr_mul(x[0], w, P);
        x[1] = fr_mul(x[1], w, P);
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += x[0].v[i] + x[1].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_frmul_cios(uint64_t *out, FrParams P, uint32_t seed)
{
    Fr x[2], w;
    for (int i = 0; i < 8; i++) { x[0].v[i] = seed + i + threadIdx.x; x[1].v[i] = seed * 3 + i + blockIdx.x; w.v[i] = seed * 7 + i; }
    x[0].v[7] &= 0x0fffffff; x[1].v[7] &= 0x0fffffff; w.v[7] &= 0x0fffffff;
    for (int it = 0; it < MUL_ITERS; it++) {
        x[0] = fr_mul_cios(x[0], w, P);
        x[1] = fr_mul_cios(x[1], w, P);
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += x[0].v[i] + x[1].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fr9mul(uint64_t *out, Fr9Params Q, uint32_t seed)
{
    Fr9 x[2], w;
    for (int i = 0; i < 9; i++) {
        x[0].v[i] = (seed + i + threadIdx.x) & HODOR_M29; x[1].v[i] = (seed * 3 + i + blockIdx.x) & HODOR_M29;
        w.v[i] = (seed * 7 + i) & HODOR_M29;
    }
    x[0].v[8] &= 0xfffff; x[1].v[8] &= 0xfffff; w.v[8] &= 0xfffff;
    for (int it = 0; it < MUL_ITERS; it++) {
        x[0] = fr9_mul(x[0], w, Q);
        x[1] = fr9_mul(x[1], w, Q);
    }
    uint64_t s = 0;
    for (int i = 0; i < 9; i++) s += x[0].v[i] + x[1].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fr9addsub(uint64_t *out, Fr9Params Q, uint32_t seed)
{
    Fr9 x[2], w;
    for (int i = 0; i < 9; i++) {
        x[0].v[i] = (seed + i + threadIdx.x) & HODOR_M29; x[1].v[i] = (seed * 3 + i + blockIdx.x) & HODOR_M29;
        w.v[i] = (seed * 7 + i) & HODOR_M29;
    }
    for (int it = 0; it < MUL_ITERS; it++) {
        x[0] = fr9_add(x[0], w);
        x[1] = fr9_sub(x[1], w, Q);
        fr9_normalize(x[0]);
        fr9_normalize(x[1]);
    }
    uint64_t s = 0;
    for (int i = 0; i < 9; i++) s += x[0].v[i] + x[1].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fraddsub(uint64_t *out, FrParams P, uint32_t seed)
{
    Fr x[2], w;
    for (int i = 0; i < 8; i++) { x[0].v[i] = seed + i + threadIdx.x; x[1].v[i] = seed * 3 + i + blockIdx.x; w.v[i] = seed * 7 + i; }
    x[0].v[7] &= 0x0fffffff; x[1].v[7] &= 0x0fffffff; w.v[7] &= 0x0fffffff;
    for (int it = 0; it < MUL_ITERS; it++) {
        x[0] = fr_add(x[0], w, P);
        x[1] = fr_sub(x[1], w, P);
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += x[0].v[i] + x[1].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// data x W3-constant product (fr9w3.cuh): 108 mads + 3 mul_lo, constant in registers
__global__ void k_fr9mul3(uint64_t *out, Fr9Params Q, uint32_t seed)
{
    Fr9 x[2];
    Fr9W3 w;
    for (int i = 0; i < 9; i++) {
        x[0].v[i] = (seed + i + threadIdx.x) & HODOR_M29; x[1].v[i] = (seed * 3 + i + blockIdx.x) & HODOR_M29;
        for (int c = 0; c < 3; c++) w.w[c][i] = (seed * (7 + c) + i) & HODOR_M29;
    }
    x[0].v[8] &= 0xfffff; x[1].v[8] &= 0xfffff;
    for (int c = 0; c < 3; c++) w.w[c][8] &= 0xfffff;
    for (int it = 0; it < MUL_ITERS; it++) {
        x[0] = fr9_mul3(x[0], w, Q);
        x[1] = fr9_mul3(x[1], w, Q);
    }
    uint64_t s = 0;
    for (int i = 0; i < 9; i++) s += x[0].v[i] + x[1].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Two W3 products by the SAME constant, column by column in lock step: two independent accumulator chains
// (does a wave need instruction-level parallelism of its own, or do the other resident waves hide the
// dependent-mad latency of the single pinned chain of fr9_mul3?)
__device__ __forceinline__ void fr9_mul3x2(const Fr9 &a0, const Fr9 &a1, const Fr9W3 &W, const Fr9Params &P,
                                           Fr9 &r0, Fr9 &r1)
{
    uint32_t m0[3], m1[3];
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (k - j >= 0 && k - j < 9) {
                    FR9_MAD(acc0, a0.v[3 * c + j], W.w[c][k - j]);
                    FR9_MAD(acc1, a1.v[3 * c + j], W.w[c][k - j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j < k && k - j < 9) {
                FR9_MAD(acc0, m0[j], P.p[k - j]);
                FR9_MAD(acc1, m1[j], P.p[k - j]);
            }
        }
        if (k < 3) {
            m0[k] = ((uint32_t)acc0 * P.pinv) & HODOR_M29;
            m1[k] = ((uint32_t)acc1 * P.pinv) & HODOR_M29;
            FR9_MAD(acc0, m0[k], P.p[0]);
            FR9_MAD(acc1, m1[k], P.p[0]);
        } else {
            r0.v[k - 3] = (uint32_t)acc0 & HODOR_M29;
            r1.v[k - 3] = (uint32_t)acc1 & HODOR_M29;
        }
        acc0 >>= 29;
        acc1 >>= 29;
    }
    r0.v[8] = (uint32_t)acc0;
    r1.v[8] = (uint32_t)acc1;
}

template <int CHAINS>
__global__ void k_fr9mul3_occ(uint64_t *out, Fr9Params Q, uint32_t seed)
{
    extern __shared__ uint32_t occupancy_pad[];   // dynamic LDS only limits the resident workgroups
    Fr9 x[2], y[2];
    Fr9W3 w;
    for (int i = 0; i < 9; i++) {
        x[0].v[i] = (seed + i + threadIdx.x) & HODOR_M29; x[1].v[i] = (seed * 3 + i + blockIdx.x) & HODOR_M29;
        for (int c = 0; c < 3; c++) w.w[c][i] = (seed * (7 + c) + i) & HODOR_M29;
    }
    x[0].v[8] &= 0xfffff; x[1].v[8] &= 0xfffff;
    for (int c = 0; c < 3; c++) w.w[c][8] &= 0xfffff;
    for (int it = 0; it < MUL_ITERS; it++) {
        if (CHAINS == 2) {
            fr9_mul3x2(x[0], x[1], w, Q, y[0], y[1]);
            x[0] = y[0]; x[1] = y[1];
        } else {
            x[0] = fr9_mul3(x[0], w, Q);
            x[1] = fr9_mul3(x[1], w, Q);
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 9; i++) s += x[0].v[i] + x[1].v[i];
    if (seed == 0xffffffffu) occupancy_pad[threadIdx.x] = (uint32_t)s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// BLAKE2s compression rate (blake2s.cuh, one hash per lane) against resident waves per SIMD: NODE = 64-byte
// message (all 16 words live), leaf = 32-byte message (words 8..15 zero).  Chained: each digest is the next
// message, so nothing can be hoisted.
constexpr int B2S_ITERS = 64;
template <bool NODE>
__global__ void k_b2s_occ(uint32_t *out, B2Mid mid, uint32_t seed)
{
    extern __shared__ uint32_t occupancy_pad[];
    uint32_t l[8], r[8], o[8];
    for (int i = 0; i < 8; i++) { l[i] = seed + i + threadIdx.x; r[i] = seed * 3 + i + blockIdx.x; }
    for (int it = 0; it < B2S_ITERS; it++) {
        if (NODE) {
            b2s_node(mid, l, r, o);
            for (int i = 0; i < 8; i++) { r[i] = l[i]; l[i] = o[i]; }
        } else {
            b2s_leaf(mid, make_uint4(l[0], l[1], l[2], l[3]), make_uint4(l[4], l[5], l[6], l[7]), o);
            for (int i = 0; i < 8; i++) l[i] = o[i];
        }
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= l[i];
    if (seed == 0xffffffffu) occupancy_pad[threadIdx.x] = x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// Instruction-mix proxy of a 5 x 52-bit double-precision-FMA Montgomery product (the scheme of Emmart,
// Zheng, Weems: hi = fma(a, b, 2^104), lo = fma(a, b, (2^104 + 2^52) - hi), bit patterns summed as
// 64-bit integers): per limb product 2 v_fma_f64 + 1 v_add_f64 + 2 64-bit integer adds, 50 limb products
// (a x b and m x p), plus 10 int->double conversions of the carried columns.  The values are not a
// real product (no rounding-mode switch, no final carry pass): only the issue cost is measured.
__global__ void k_dpfma_proxy(uint64_t *out, uint32_t a0, uint32_t b0)
{
    double a[5], b[5], p[5];
    for (int i = 0; i < 5; i++) { a[i] = (double)(a0 + i + threadIdx.x); b[i] = (double)(b0 + 3 * i + 1); p[i] = (double)(7 * i + 5); }
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    uint64_t col[10];
    for (int it = 0; it < MUL_ITERS; it++) {
#pragma unroll
        for (int k = 0; k < 10; k++) col[k] = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int i = 0; i < 5; i++) {
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    double y = half ? p[j] : b[j];
                    double hi = __builtin_fma(a[i], y, C1);
                    double lo = __builtin_fma(a[i], y, C2 - hi);
                    col[i + j + 1] += (uint64_t)__double_as_longlong(hi);
                    col[i + j] += (uint64_t)__double_as_longlong(lo);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) a[i] = (double)(uint32_t)(col[i] ^ col[i + 5]) * 0x1p-8 + 1.0;
    }
    uint64_t s = 0;
    for (int i = 0; i < 10; i++) s += col[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Memory pattern of a two-pass 2^12 x 2^12 plan for the 2^24-point transform (one 4096-point
// sub-transform per workgroup, C = 1): a workgroup gathers 4096 elements that lie 4096 elements apart
// (32-byte pieces), parks them in LDS, and writes them back either contiguously (first pass) or with the
// same stride (second pass).  xcd_aware: the four workgroups that share each 128-byte line are mapped
// to the same XCD (block b runs on XCD b % 8) and dispatched back to back.
__global__ void __launch_bounds__(1024)
k_stride_probe(const uint4 *src, uint4 *dst, int strided_store, int xcd_aware)
{
    extern __shared__ uint4 lds[];
    uint32_t b = blockIdx.x, j;
    if (xcd_aware) {
        uint32_t x = b & 7, q = b >> 3;
        j = ((q >> 2) << 5) + (x << 2) + (q & 3);
    } else {
        j = b;
    }
    for (uint32_t i = threadIdx.x; i < 4096; i += 1024) {
        const uint4 *s = src + 2 * ((uint64_t)j + ((uint64_t)i << 12));
        lds[i] = s[0];
        lds[4096 + i] = s[1];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096; i += 1024) {
        uint32_t r = (i * 2654435761u) >> 20;   // some permutation-ish shuffle so LDS is really used
        r = (r & ~4095u) | i;
        uint4 *d = strided_store ? dst + 2 * ((uint64_t)j + ((uint64_t)i << 12)) : dst + 2 * (((uint64_t)j << 12) + i);
        d[0] = lds[r & 4095];
        d[1] = lds[4096 + (r & 4095)];
    }
}

// The pass kernel's read pattern in isolation: a workgroup of 256 threads gathers a 1024-element tile =
// 256 rows of 128 contiguous bytes, the rows `row_stride` bytes apart (n/R elements = 2 MiB at 2^24, R = 256),
// four elements per thread all requested before the first is used, and writes the 32 KiB contiguously.
// 16384 workgroups cover 512 MiB.  Varying the stride by a few hundred bytes separates DRAM channel /
// bank aliasing of the power-of-two stride from everything else.
__global__ void __launch_bounds__(256)
k_row_gather(const uint4 *src, uint4 *dst, uint64_t row_stride_bytes)
{
    const uint32_t tid = threadIdx.x;
    const char *base = reinterpret_cast<const char *>(src) + (uint64_t)blockIdx.x * 128;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t e = tid + 256 * k, c = e & 3, i = e >> 2;
        const uint4 *p = reinterpret_cast<const uint4 *>(base + (uint64_t)i * row_stride_bytes + c * 32);
        v[2 * k] = p[0];
        v[2 * k + 1] = p[1];
    }
    uint4 *out = dst + 2 * ((uint64_t)blockIdx.x * 1024);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        out[2 * (tid + 256 * k)] = v[2 * k];
        out[2 * (tid + 256 * k) + 1] = v[2 * k + 1];
    }
}

__global__ void k_copy(const uint4 *in, uint4 *out, size_t n)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

template <class F>
static float time_it(F f, int reps = 5)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device: %s CUs=%d clock=%d MHz lds/block=%zu maxLds/CU=%zu\n", prop.name, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    uint64_t *out;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    double lanes = (double)blocks * threads;
#define RUN(name, kern)                                                                               \
    {                                                                                                 \
        float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); }); \
        double ops = lanes * ITERS * ILP;                                                             \
        printf("%-16s %8.3f ms  %8.2f Gop/s  %6.2f lane-ops/clk/CU @2.4GHz\n", name, ms, ops / ms * 1e-6, \
               ops / (ms * 1e-3) / (prop.multiProcessorCount * 2.4e9));                               \
    }
    RUN("v_mad_u64_u32", k_mad64)
    RUN("v_mul_lo_u32", k_mullo)
    RUN("v_mul_hi_u32", k_mulhi)
    RUN("v_mad_u32_u24", k_mad24)
    RUN("v_addc_co_u32", k_addc)
    RUN("v_add_u32", k_add32)
    RUN("v_lshl_add_u64", k_lshladd64)
    RUN("v_fma_f64", k_fma64)
    RUN("v_fma_f32", k_fma32)
    RUN("v_alignbit_b32", k_alignbit)
    RUN("v_xor_b32", k_xor)
    RUN("v_add3_u32", k_add3)
    RUN("v_perm_b32", k_perm)
    RUN("v_and_b32", k_lshrrev64x)
    RUN("v_lshrrev_b64", k_lshr64)
    RUN("v_alignbit_b32", k_alignbit2)
    RUN("v_xor_b32_sdwa", k_xor_sdwa)
    RUN("rot16: 2x sdwa", k_rot16_sdwa)
    RUN("rot16: xor+align", k_rot16_plain)
    RUN("rot16: distance 2", k_rot16_dist2)
    RUN("rot16: grouped", k_rot16_grouped)
    RUN("v_xad_u32", k_xad)
    RUN("v_xor_b32_dpp", k_xor_dpp)
    RUN("v_add_u32_dpp", k_add_dpp)
    RUN("v_pk_add_u16 opsel", k_pk_swap)
    RUN("v_add_u32_e64", k_add_e64)
    RUN("v_xor_b32_e64", k_xor_e64)
    RUN("v_pk_add_u16", k_pk_add)
    RUN("v_lshl_or_b32", k_lshl_or)
    RUN("v_bfi_b32", k_bfi)
    FrParams P;
    // BLS12-381 Fr (the field in src/bn256.rs)
    const uint32_t p[8] = {0x00000001, 0xffffffff, 0xfffe5bfe, 0x53bda402, 0x09a1d805, 0x3339d808, 0x299d7d48, 0x73eda753};
    for (int i = 0; i < 8; i++) { P.p[i] = p[i]; P.one[i] = 0; }
    P.pinv = 0xffffffff;
    {
        float ms = time_it([&] { hipLaunchKernelGGL(k_frmul, dim3(blocks), dim3(threads), 0, 0, out, P, 12345u); });
        double muls = lanes * MUL_ITERS * 2;
        printf("%-16s %8.3f ms  %8.2f Gmul/s\n", "fr_mul (FIPS)", ms, muls / ms * 1e-6);
        ms = time_it([&] { hipLaunchKernelGGL(k_frmul_cios, dim3(blocks), dim3(threads), 0, 0, out, P, 12345u); });
        printf("%-16s %8.3f ms  %8.2f Gmul/s\n", "fr_mul_cios", ms, muls / ms * 1e-6);
        ms = time_it([&] { hipLaunchKernelGGL(k_fraddsub, dim3(blocks), dim3(threads), 0, 0, out, P, 12345u); });
        printf("%-16s %8.3f ms  %8.2f Gop/s\n", "fr_add+fr_sub", ms, muls / ms * 1e-6);
        Fr9Params Q;
        const uint32_t p9[9] = {0x00000001, 0x1ffffff8, 0x1f96ffbf, 0x1b4805ff, 0x0c0a77b4, 0x0c0404d0, 0x11f52199, 0x1a94ce9d, 0x0073eda7};
        for (int i = 0; i < 9; i++) { Q.p[i] = p9[i]; Q.c4p[i] = 0x20000000u + p9[i]; }
        Q.pinv = 0x1fffffff; Q.mu = 2262; Q.red_shift = 18;
        ms = time_it([&] { hipLaunchKernelGGL(k_fr9mul, dim3(blocks), dim3(threads), 0, 0, out, Q, 12345u); });
        printf("%-16s %8.3f ms  %8.2f Gmul/s\n", "fr9_mul (9x29)", ms, muls / ms * 1e-6);
        ms = time_it([&] { hipLaunchKernelGGL(k_fr9mul3, dim3(blocks), dim3(threads), 0, 0, out, Q, 12345u); });
        printf("%-16s %8.3f ms  %8.2f Gmul/s\n", "fr9_mul3 (W3)", ms, muls / ms * 1e-6);
        // the product rate against resident waves per SIMD (256-thread workgroups, dynamic LDS as the limiter):
        // the pass kernel runs at 4; one pinned chain (fr9_mul3) against two interleaved chains
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_fr9mul3_occ<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_fr9mul3_occ<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int wgs = 1; wgs <= 8; wgs *= 2) {
            size_t lds = wgs == 8 ? 0 : (size_t)(160 * 1024 / wgs) - 1024;
            ms = time_it([&] { hipLaunchKernelGGL(k_fr9mul3_occ<1>, dim3(blocks), dim3(threads), lds, 0, out, Q, 12345u); });
            float ms2 = time_it([&] { hipLaunchKernelGGL(k_fr9mul3_occ<2>, dim3(blocks), dim3(threads), lds, 0, out, Q, 12345u); });
            printf("fr9_mul3 at %d waves/SIMD: 1 chain %8.2f Gmul/s   2 chains %8.2f Gmul/s\n", wgs, muls / ms * 1e-6, muls / ms2 * 1e-6);
        }
        ms = time_it([&] { hipLaunchKernelGGL(k_dpfma_proxy, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u); });
        printf("%-16s %8.3f ms  %8.2f Gmul/s (issue-cost proxy, 5x52-bit DP-FMA Montgomery: 100 fma + 50 add_f64 + 100 add_u64)\n", "dp-fma proxy", ms, lanes * MUL_ITERS / ms * 1e-6);
        ms = time_it([&] { hipLaunchKernelGGL(k_fr9addsub, dim3(blocks), dim3(threads), 0, 0, out, Q, 12345u); });
        printf("%-16s %8.3f ms  %8.2f Gop/s (incl. normalize)\n", "fr9_add+sub", ms, muls / ms * 1e-6);
    }
    {
        B2Mid mid;
        for (int i = 0; i < 8; i++) mid.h[i] = 0x9e3779b9u * (i + 1);
        uint32_t *o32;
        hipMalloc(&o32, (size_t)4096 * 256 * 4);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_b2s_occ<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_b2s_occ<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const double compr = 4096.0 * 256 * B2S_ITERS;
        const int occs[] = {1, 2, 4, 6, 8};
        for (int wgs : occs) {
            size_t lds = wgs == 8 ? 0 : (size_t)(160 * 1024 / wgs) - 1024;
            float ms = time_it([&] { hipLaunchKernelGGL(k_b2s_occ<true>, dim3(4096), dim3(256), lds, 0, o32, mid, 12345u); });
            float ms2 = time_it([&] { hipLaunchKernelGGL(k_b2s_occ<false>, dim3(4096), dim3(256), lds, 0, o32, mid, 12345u); });
            printf("blake2s compression at %d waves/SIMD: node %8.2f G/s (%5.1f ps)   leaf %8.2f G/s (%5.1f ps)\n", wgs,
                   compr / ms * 1e-6, ms * 1e9 / compr, compr / ms2 * 1e-6, ms2 * 1e9 / compr);
        }
        hipFree(o32);
    }
    {
        size_t n = (size_t)1 << 26;   // 1 GiB of uint4
        uint4 *a, *b;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
        hipMemset(a, 1, n * 16);
        float ms = time_it([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(threads), 0, 0, a, b, n); });
        printf("%-16s %8.3f ms  %8.2f GB/s (read+write)\n", "copy 1GiB", ms, 2.0 * n * 16 / ms * 1e-6);
        // the pass kernel's gather against the row stride (bytes): 2 MiB = n/R rows of the 2^24 transform
        {
            const uint64_t strides[] = {2097152, 2097152 + 128, 2097152 + 256, 2097152 + 512, 2097152 + 4096, 2097152 + 65536,
                                        1048576, 524288};
            for (uint64_t st : strides) {
                ms = time_it([&] { hipLaunchKernelGGL(k_row_gather, dim3(16384), dim3(256), 0, 0, a, b, st); });
                printf("row gather, stride %8llu B (2 MiB %+8lld): %8.3f ms  %8.2f GB/s (read+write)\n", (unsigned long long)st,
                       (long long)st - 2097152, ms, 2.0 * 512 * 1048576 / ms * 1e-6);
            }
        }
        // 2^24 elements of 32 B = 512 MiB: the two-pass plan's memory patterns
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_stride_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int strided = 0; strided < 2; strided++)
            for (int xcd = 0; xcd < 2; xcd++) {
                ms = time_it([&] { hipLaunchKernelGGL(k_stride_probe, dim3(4096), dim3(1024), 128 * 1024, 0, a, b, strided, xcd); });
                printf("stride probe (%s store, %s)  %8.3f ms  %8.2f GB/s\n", strided ? "strided" : "contiguous",
                       xcd ? "xcd-aware" : "plain order", ms, 2.0 * 512 * 1048576 / ms * 1e-6);
            }
    }
    return 0;
}
