// dev tool: issue cost of the integer instructions the seed-index kernel is made of (MI355X, 4 waves per SIMD).
// Each kernel runs ITER iterations of an unrolled chain-free block of N copies of one instruction on independent
// registers; time / (ITER * N * waves per SIMD) = cycles per wave instruction at the clock the chip holds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define ITER 4096
typedef unsigned int u32;
typedef unsigned long long u64;

#define KERNEL(name, decl, body)                                                            \
  __global__ void __launch_bounds__(1024, 1) name(u32* out, u32 seed) {                     \
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
    decl;                                                                                   \
    for (int it = 0; it < ITER; ++it) {                                                     \
      body body body body                                                                   \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
  }
// one "body" = 8 instructions
#define OP8(ins) asm volatile(ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define I_ADD(n) "v_add_u32 %" #n ", 1, %" #n "\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 2, %" #n "\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", 63, %" #n "\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 13\n"
#define I_BFEV(n) "v_bfe_u32 %" #n ", %" #n ", %" #n ", 1\n"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 21\n"
#define I_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %" #n ", %" #n " bitop3:0x96\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %" #n "\n"
#define I_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %" #n ", %" #n "\n"
#define I_FFBL(n) "v_ffbl_b32 %" #n ", %" #n "\n"
#define I_OR3(n) "v_or3_b32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %" #n ", vcc\n"
#define I_CMP(n) "v_cmp_ne_u32 vcc, 0, %" #n "\n"
#define I_DPPQ(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_DPPR(n) "v_mov_b32_dpp %" #n ", %" #n " row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define I_MUL(n) "v_mul_lo_u32 %" #n ", %" #n ", %" #n "\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %" #n ", %" #n "\n"
KERNEL(k_add, , OP8(I_ADD))
KERNEL(k_lshl_or, , OP8(I_LSHLOR))
KERNEL(k_and_or, , OP8(I_ANDOR))
KERNEL(k_bfe_const, , OP8(I_BFE))
KERNEL(k_bfe_var, , OP8(I_BFEV))
KERNEL(k_alignbit, , OP8(I_ALIGN))
KERNEL(k_bitop3, , OP8(I_BITOP3))
KERNEL(k_perm, , OP8(I_PERM))
KERNEL(k_bcnt, , OP8(I_BCNT))
KERNEL(k_mbcnt, , OP8(I_MBCNT))
KERNEL(k_ffbl, , OP8(I_FFBL))
KERNEL(k_or3, , OP8(I_OR3))
KERNEL(k_cndmask, , OP8(I_CNDMASK))
KERNEL(k_cmp, , OP8(I_CMP))
KERNEL(k_dpp_quad, , OP8(I_DPPQ))
KERNEL(k_dpp_ror8, , OP8(I_DPPR))
KERNEL(k_mul_lo, , OP8(I_MUL))
KERNEL(k_mad24, , OP8(I_MAD24))

// 64-bit shifts: 4 register pairs
#define KERNEL64(name, ins)                                                                 \
  __global__ void __launch_bounds__(1024, 1) name(u32* out, u32 seed) {                     \
    u64 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u; u32 s = (seed & 3u) + 1u; \
    for (int it = 0; it < ITER; ++it) {                                                     \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
      asm volatile(ins : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));                  \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(a0 ^ a1 ^ a2 ^ a3);                  \
  }
KERNEL64(k_lshl64_const, "v_lshlrev_b64 %0, 2, %0\nv_lshlrev_b64 %1, 2, %1\nv_lshlrev_b64 %2, 2, %2\nv_lshlrev_b64 %3, 2, %3\n")
KERNEL64(k_lshr64_var, "v_lshrrev_b64 %0, %4, %0\nv_lshrrev_b64 %1, %4, %1\nv_lshrrev_b64 %2, %4, %2\nv_lshrrev_b64 %3, %4, %3\n")
KERNEL64(k_add64, "v_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %1, %1, 0, %2\nv_lshl_add_u64 %2, %2, 0, %3\nv_lshl_add_u64 %3, %3, 0, %0\n")

// LDS-pipe instructions: random byte reads, swizzles
__global__ void __launch_bounds__(1024, 1) k_ds_read_u8_random(u32* out, u32 seed) {
  __shared__ unsigned char tab[65536];
  for (u32 i = threadIdx.x; i < 65536u / 4u; i += 1024u) reinterpret_cast<u32*>(tab)[i] = i * 2654435761u;
  __syncthreads();
  u32 x = threadIdx.x * 2654435761u + seed, acc = 0;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; acc += tab[x >> 16]; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(1024, 1) k_ds_read_u8_samebank(u32* out, u32 seed) { // conflict-free: lane l reads bank l
  __shared__ unsigned char tab[65536];
  for (u32 i = threadIdx.x; i < 65536u / 4u; i += 1024u) reinterpret_cast<u32*>(tab)[i] = i * 2654435761u;
  __syncthreads();
  u32 x = threadIdx.x * 2654435761u + seed, acc = 0;
  const u32 lane4 = (threadIdx.x & 31u) * 4u;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; acc += tab[((x >> 16) & 0xFF80u) | lane4]; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(1024, 1) k_ds_swizzle(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      a0 = __builtin_amdgcn_ds_swizzle(a0, 0x401F); // xor 16 within 32 lanes
      a1 = __builtin_amdgcn_ds_swizzle(a1, 0x201F);
      a2 = __builtin_amdgcn_ds_swizzle(a2, 0x101F);
      a3 = __builtin_amdgcn_ds_swizzle(a3, 0x081F);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
__global__ void __launch_bounds__(1024, 1) k_ds_bpermute(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
  const u32 idx = ((threadIdx.x + 1u) & 63u) * 4u;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      a0 = __builtin_amdgcn_ds_bpermute(idx, a0); a1 = __builtin_amdgcn_ds_bpermute(idx, a1);
      a2 = __builtin_amdgcn_ds_bpermute(idx, a2); a3 = __builtin_amdgcn_ds_bpermute(idx, a3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
__global__ void __launch_bounds__(1024, 1) k_permlane32_swap(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\n"
                 "v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\nv_permlane16_swap_b32 %6, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_and2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_and_b32 %0, %0, %1\n"
  "v_and_b32 %1, %1, %2\n"
  "v_and_b32 %2, %2, %3\n"
  "v_and_b32 %3, %3, %4\n"
  "v_and_b32 %4, %4, %5\n"
  "v_and_b32 %5, %5, %6\n"
  "v_and_b32 %6, %6, %7\n"
  "v_and_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, %0, %1\n"
  "v_and_b32 %1, %1, %2\n"
  "v_and_b32 %2, %2, %3\n"
  "v_and_b32 %3, %3, %4\n"
  "v_and_b32 %4, %4, %5\n"
  "v_and_b32 %5, %5, %6\n"
  "v_and_b32 %6, %6, %7\n"
  "v_and_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, %0, %1\n"
  "v_and_b32 %1, %1, %2\n"
  "v_and_b32 %2, %2, %3\n"
  "v_and_b32 %3, %3, %4\n"
  "v_and_b32 %4, %4, %5\n"
  "v_and_b32 %5, %5, %6\n"
  "v_and_b32 %6, %6, %7\n"
  "v_and_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, %0, %1\n"
  "v_and_b32 %1, %1, %2\n"
  "v_and_b32 %2, %2, %3\n"
  "v_and_b32 %3, %3, %4\n"
  "v_and_b32 %4, %4, %5\n"
  "v_and_b32 %5, %5, %6\n"
  "v_and_b32 %6, %6, %7\n"
  "v_and_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_or2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_or_b32 %0, %0, %1\n"
  "v_or_b32 %1, %1, %2\n"
  "v_or_b32 %2, %2, %3\n"
  "v_or_b32 %3, %3, %4\n"
  "v_or_b32 %4, %4, %5\n"
  "v_or_b32 %5, %5, %6\n"
  "v_or_b32 %6, %6, %7\n"
  "v_or_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_or_b32 %0, %0, %1\n"
  "v_or_b32 %1, %1, %2\n"
  "v_or_b32 %2, %2, %3\n"
  "v_or_b32 %3, %3, %4\n"
  "v_or_b32 %4, %4, %5\n"
  "v_or_b32 %5, %5, %6\n"
  "v_or_b32 %6, %6, %7\n"
  "v_or_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_or_b32 %0, %0, %1\n"
  "v_or_b32 %1, %1, %2\n"
  "v_or_b32 %2, %2, %3\n"
  "v_or_b32 %3, %3, %4\n"
  "v_or_b32 %4, %4, %5\n"
  "v_or_b32 %5, %5, %6\n"
  "v_or_b32 %6, %6, %7\n"
  "v_or_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_or_b32 %0, %0, %1\n"
  "v_or_b32 %1, %1, %2\n"
  "v_or_b32 %2, %2, %3\n"
  "v_or_b32 %3, %3, %4\n"
  "v_or_b32 %4, %4, %5\n"
  "v_or_b32 %5, %5, %6\n"
  "v_or_b32 %6, %6, %7\n"
  "v_or_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_xor2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_xor_b32 %0, %0, %1\n"
  "v_xor_b32 %1, %1, %2\n"
  "v_xor_b32 %2, %2, %3\n"
  "v_xor_b32 %3, %3, %4\n"
  "v_xor_b32 %4, %4, %5\n"
  "v_xor_b32 %5, %5, %6\n"
  "v_xor_b32 %6, %6, %7\n"
  "v_xor_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_xor_b32 %0, %0, %1\n"
  "v_xor_b32 %1, %1, %2\n"
  "v_xor_b32 %2, %2, %3\n"
  "v_xor_b32 %3, %3, %4\n"
  "v_xor_b32 %4, %4, %5\n"
  "v_xor_b32 %5, %5, %6\n"
  "v_xor_b32 %6, %6, %7\n"
  "v_xor_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_xor_b32 %0, %0, %1\n"
  "v_xor_b32 %1, %1, %2\n"
  "v_xor_b32 %2, %2, %3\n"
  "v_xor_b32 %3, %3, %4\n"
  "v_xor_b32 %4, %4, %5\n"
  "v_xor_b32 %5, %5, %6\n"
  "v_xor_b32 %6, %6, %7\n"
  "v_xor_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_xor_b32 %0, %0, %1\n"
  "v_xor_b32 %1, %1, %2\n"
  "v_xor_b32 %2, %2, %3\n"
  "v_xor_b32 %3, %3, %4\n"
  "v_xor_b32 %4, %4, %5\n"
  "v_xor_b32 %5, %5, %6\n"
  "v_xor_b32 %6, %6, %7\n"
  "v_xor_b32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_andc(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_and_b32 %0, 63, %0\n"
  "v_and_b32 %1, 63, %1\n"
  "v_and_b32 %2, 63, %2\n"
  "v_and_b32 %3, 63, %3\n"
  "v_and_b32 %4, 63, %4\n"
  "v_and_b32 %5, 63, %5\n"
  "v_and_b32 %6, 63, %6\n"
  "v_and_b32 %7, 63, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, 63, %0\n"
  "v_and_b32 %1, 63, %1\n"
  "v_and_b32 %2, 63, %2\n"
  "v_and_b32 %3, 63, %3\n"
  "v_and_b32 %4, 63, %4\n"
  "v_and_b32 %5, 63, %5\n"
  "v_and_b32 %6, 63, %6\n"
  "v_and_b32 %7, 63, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, 63, %0\n"
  "v_and_b32 %1, 63, %1\n"
  "v_and_b32 %2, 63, %2\n"
  "v_and_b32 %3, 63, %3\n"
  "v_and_b32 %4, 63, %4\n"
  "v_and_b32 %5, 63, %5\n"
  "v_and_b32 %6, 63, %6\n"
  "v_and_b32 %7, 63, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, 63, %0\n"
  "v_and_b32 %1, 63, %1\n"
  "v_and_b32 %2, 63, %2\n"
  "v_and_b32 %3, 63, %3\n"
  "v_and_b32 %4, 63, %4\n"
  "v_and_b32 %5, 63, %5\n"
  "v_and_b32 %6, 63, %6\n"
  "v_and_b32 %7, 63, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_andlit(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_and_b32 %0, 0x1ff8, %0\n"
  "v_and_b32 %1, 0x1ff8, %1\n"
  "v_and_b32 %2, 0x1ff8, %2\n"
  "v_and_b32 %3, 0x1ff8, %3\n"
  "v_and_b32 %4, 0x1ff8, %4\n"
  "v_and_b32 %5, 0x1ff8, %5\n"
  "v_and_b32 %6, 0x1ff8, %6\n"
  "v_and_b32 %7, 0x1ff8, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, 0x1ff8, %0\n"
  "v_and_b32 %1, 0x1ff8, %1\n"
  "v_and_b32 %2, 0x1ff8, %2\n"
  "v_and_b32 %3, 0x1ff8, %3\n"
  "v_and_b32 %4, 0x1ff8, %4\n"
  "v_and_b32 %5, 0x1ff8, %5\n"
  "v_and_b32 %6, 0x1ff8, %6\n"
  "v_and_b32 %7, 0x1ff8, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, 0x1ff8, %0\n"
  "v_and_b32 %1, 0x1ff8, %1\n"
  "v_and_b32 %2, 0x1ff8, %2\n"
  "v_and_b32 %3, 0x1ff8, %3\n"
  "v_and_b32 %4, 0x1ff8, %4\n"
  "v_and_b32 %5, 0x1ff8, %5\n"
  "v_and_b32 %6, 0x1ff8, %6\n"
  "v_and_b32 %7, 0x1ff8, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32 %0, 0x1ff8, %0\n"
  "v_and_b32 %1, 0x1ff8, %1\n"
  "v_and_b32 %2, 0x1ff8, %2\n"
  "v_and_b32 %3, 0x1ff8, %3\n"
  "v_and_b32 %4, 0x1ff8, %4\n"
  "v_and_b32 %5, 0x1ff8, %5\n"
  "v_and_b32 %6, 0x1ff8, %6\n"
  "v_and_b32 %7, 0x1ff8, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_lshlc(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_lshlrev_b32 %0, 2, %0\n"
  "v_lshlrev_b32 %1, 2, %1\n"
  "v_lshlrev_b32 %2, 2, %2\n"
  "v_lshlrev_b32 %3, 2, %3\n"
  "v_lshlrev_b32 %4, 2, %4\n"
  "v_lshlrev_b32 %5, 2, %5\n"
  "v_lshlrev_b32 %6, 2, %6\n"
  "v_lshlrev_b32 %7, 2, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshlrev_b32 %0, 2, %0\n"
  "v_lshlrev_b32 %1, 2, %1\n"
  "v_lshlrev_b32 %2, 2, %2\n"
  "v_lshlrev_b32 %3, 2, %3\n"
  "v_lshlrev_b32 %4, 2, %4\n"
  "v_lshlrev_b32 %5, 2, %5\n"
  "v_lshlrev_b32 %6, 2, %6\n"
  "v_lshlrev_b32 %7, 2, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshlrev_b32 %0, 2, %0\n"
  "v_lshlrev_b32 %1, 2, %1\n"
  "v_lshlrev_b32 %2, 2, %2\n"
  "v_lshlrev_b32 %3, 2, %3\n"
  "v_lshlrev_b32 %4, 2, %4\n"
  "v_lshlrev_b32 %5, 2, %5\n"
  "v_lshlrev_b32 %6, 2, %6\n"
  "v_lshlrev_b32 %7, 2, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshlrev_b32 %0, 2, %0\n"
  "v_lshlrev_b32 %1, 2, %1\n"
  "v_lshlrev_b32 %2, 2, %2\n"
  "v_lshlrev_b32 %3, 2, %3\n"
  "v_lshlrev_b32 %4, 2, %4\n"
  "v_lshlrev_b32 %5, 2, %5\n"
  "v_lshlrev_b32 %6, 2, %6\n"
  "v_lshlrev_b32 %7, 2, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_lshrc(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_lshrrev_b32 %0, 13, %0\n"
  "v_lshrrev_b32 %1, 13, %1\n"
  "v_lshrrev_b32 %2, 13, %2\n"
  "v_lshrrev_b32 %3, 13, %3\n"
  "v_lshrrev_b32 %4, 13, %4\n"
  "v_lshrrev_b32 %5, 13, %5\n"
  "v_lshrrev_b32 %6, 13, %6\n"
  "v_lshrrev_b32 %7, 13, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshrrev_b32 %0, 13, %0\n"
  "v_lshrrev_b32 %1, 13, %1\n"
  "v_lshrrev_b32 %2, 13, %2\n"
  "v_lshrrev_b32 %3, 13, %3\n"
  "v_lshrrev_b32 %4, 13, %4\n"
  "v_lshrrev_b32 %5, 13, %5\n"
  "v_lshrrev_b32 %6, 13, %6\n"
  "v_lshrrev_b32 %7, 13, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshrrev_b32 %0, 13, %0\n"
  "v_lshrrev_b32 %1, 13, %1\n"
  "v_lshrrev_b32 %2, 13, %2\n"
  "v_lshrrev_b32 %3, 13, %3\n"
  "v_lshrrev_b32 %4, 13, %4\n"
  "v_lshrrev_b32 %5, 13, %5\n"
  "v_lshrrev_b32 %6, 13, %6\n"
  "v_lshrrev_b32 %7, 13, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshrrev_b32 %0, 13, %0\n"
  "v_lshrrev_b32 %1, 13, %1\n"
  "v_lshrrev_b32 %2, 13, %2\n"
  "v_lshrrev_b32 %3, 13, %3\n"
  "v_lshrrev_b32 %4, 13, %4\n"
  "v_lshrrev_b32 %5, 13, %5\n"
  "v_lshrrev_b32 %6, 13, %6\n"
  "v_lshrrev_b32 %7, 13, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_lshrv(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_lshrrev_b32 %0, %1, %0\n"
  "v_lshrrev_b32 %1, %2, %1\n"
  "v_lshrrev_b32 %2, %3, %2\n"
  "v_lshrrev_b32 %3, %4, %3\n"
  "v_lshrrev_b32 %4, %5, %4\n"
  "v_lshrrev_b32 %5, %6, %5\n"
  "v_lshrrev_b32 %6, %7, %6\n"
  "v_lshrrev_b32 %7, %0, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshrrev_b32 %0, %1, %0\n"
  "v_lshrrev_b32 %1, %2, %1\n"
  "v_lshrrev_b32 %2, %3, %2\n"
  "v_lshrrev_b32 %3, %4, %3\n"
  "v_lshrrev_b32 %4, %5, %4\n"
  "v_lshrrev_b32 %5, %6, %5\n"
  "v_lshrrev_b32 %6, %7, %6\n"
  "v_lshrrev_b32 %7, %0, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshrrev_b32 %0, %1, %0\n"
  "v_lshrrev_b32 %1, %2, %1\n"
  "v_lshrrev_b32 %2, %3, %2\n"
  "v_lshrrev_b32 %3, %4, %3\n"
  "v_lshrrev_b32 %4, %5, %4\n"
  "v_lshrrev_b32 %5, %6, %5\n"
  "v_lshrrev_b32 %6, %7, %6\n"
  "v_lshrrev_b32 %7, %0, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshrrev_b32 %0, %1, %0\n"
  "v_lshrrev_b32 %1, %2, %1\n"
  "v_lshrrev_b32 %2, %3, %2\n"
  "v_lshrrev_b32 %3, %4, %3\n"
  "v_lshrrev_b32 %4, %5, %4\n"
  "v_lshrrev_b32 %5, %6, %5\n"
  "v_lshrrev_b32 %6, %7, %6\n"
  "v_lshrrev_b32 %7, %0, %7\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_mov(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_mov_b32 %0, %1\n"
  "v_mov_b32 %1, %2\n"
  "v_mov_b32 %2, %3\n"
  "v_mov_b32 %3, %4\n"
  "v_mov_b32 %4, %5\n"
  "v_mov_b32 %5, %6\n"
  "v_mov_b32 %6, %7\n"
  "v_mov_b32 %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_mov_b32 %0, %1\n"
  "v_mov_b32 %1, %2\n"
  "v_mov_b32 %2, %3\n"
  "v_mov_b32 %3, %4\n"
  "v_mov_b32 %4, %5\n"
  "v_mov_b32 %5, %6\n"
  "v_mov_b32 %6, %7\n"
  "v_mov_b32 %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_mov_b32 %0, %1\n"
  "v_mov_b32 %1, %2\n"
  "v_mov_b32 %2, %3\n"
  "v_mov_b32 %3, %4\n"
  "v_mov_b32 %4, %5\n"
  "v_mov_b32 %5, %6\n"
  "v_mov_b32 %6, %7\n"
  "v_mov_b32 %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_mov_b32 %0, %1\n"
  "v_mov_b32 %1, %2\n"
  "v_mov_b32 %2, %3\n"
  "v_mov_b32 %3, %4\n"
  "v_mov_b32 %4, %5\n"
  "v_mov_b32 %5, %6\n"
  "v_mov_b32 %6, %7\n"
  "v_mov_b32 %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_add2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_add_u32 %0, %0, %1\n"
  "v_add_u32 %1, %1, %2\n"
  "v_add_u32 %2, %2, %3\n"
  "v_add_u32 %3, %3, %4\n"
  "v_add_u32 %4, %4, %5\n"
  "v_add_u32 %5, %5, %6\n"
  "v_add_u32 %6, %6, %7\n"
  "v_add_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_add_u32 %0, %0, %1\n"
  "v_add_u32 %1, %1, %2\n"
  "v_add_u32 %2, %2, %3\n"
  "v_add_u32 %3, %3, %4\n"
  "v_add_u32 %4, %4, %5\n"
  "v_add_u32 %5, %5, %6\n"
  "v_add_u32 %6, %6, %7\n"
  "v_add_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_add_u32 %0, %0, %1\n"
  "v_add_u32 %1, %1, %2\n"
  "v_add_u32 %2, %2, %3\n"
  "v_add_u32 %3, %3, %4\n"
  "v_add_u32 %4, %4, %5\n"
  "v_add_u32 %5, %5, %6\n"
  "v_add_u32 %6, %6, %7\n"
  "v_add_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_add_u32 %0, %0, %1\n"
  "v_add_u32 %1, %1, %2\n"
  "v_add_u32 %2, %2, %3\n"
  "v_add_u32 %3, %3, %4\n"
  "v_add_u32 %4, %4, %5\n"
  "v_add_u32 %5, %5, %6\n"
  "v_add_u32 %6, %6, %7\n"
  "v_add_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_sub2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_sub_u32 %0, %0, %1\n"
  "v_sub_u32 %1, %1, %2\n"
  "v_sub_u32 %2, %2, %3\n"
  "v_sub_u32 %3, %3, %4\n"
  "v_sub_u32 %4, %4, %5\n"
  "v_sub_u32 %5, %5, %6\n"
  "v_sub_u32 %6, %6, %7\n"
  "v_sub_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_sub_u32 %0, %0, %1\n"
  "v_sub_u32 %1, %1, %2\n"
  "v_sub_u32 %2, %2, %3\n"
  "v_sub_u32 %3, %3, %4\n"
  "v_sub_u32 %4, %4, %5\n"
  "v_sub_u32 %5, %5, %6\n"
  "v_sub_u32 %6, %6, %7\n"
  "v_sub_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_sub_u32 %0, %0, %1\n"
  "v_sub_u32 %1, %1, %2\n"
  "v_sub_u32 %2, %2, %3\n"
  "v_sub_u32 %3, %3, %4\n"
  "v_sub_u32 %4, %4, %5\n"
  "v_sub_u32 %5, %5, %6\n"
  "v_sub_u32 %6, %6, %7\n"
  "v_sub_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_sub_u32 %0, %0, %1\n"
  "v_sub_u32 %1, %1, %2\n"
  "v_sub_u32 %2, %2, %3\n"
  "v_sub_u32 %3, %3, %4\n"
  "v_sub_u32 %4, %4, %5\n"
  "v_sub_u32 %5, %5, %6\n"
  "v_sub_u32 %6, %6, %7\n"
  "v_sub_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_add3(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_add3_u32 %0, %0, %1, %2\n"
  "v_add3_u32 %1, %1, %2, %3\n"
  "v_add3_u32 %2, %2, %3, %4\n"
  "v_add3_u32 %3, %3, %4, %5\n"
  "v_add3_u32 %4, %4, %5, %6\n"
  "v_add3_u32 %5, %5, %6, %7\n"
  "v_add3_u32 %6, %6, %7, %0\n"
  "v_add3_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_add3_u32 %0, %0, %1, %2\n"
  "v_add3_u32 %1, %1, %2, %3\n"
  "v_add3_u32 %2, %2, %3, %4\n"
  "v_add3_u32 %3, %3, %4, %5\n"
  "v_add3_u32 %4, %4, %5, %6\n"
  "v_add3_u32 %5, %5, %6, %7\n"
  "v_add3_u32 %6, %6, %7, %0\n"
  "v_add3_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_add3_u32 %0, %0, %1, %2\n"
  "v_add3_u32 %1, %1, %2, %3\n"
  "v_add3_u32 %2, %2, %3, %4\n"
  "v_add3_u32 %3, %3, %4, %5\n"
  "v_add3_u32 %4, %4, %5, %6\n"
  "v_add3_u32 %5, %5, %6, %7\n"
  "v_add3_u32 %6, %6, %7, %0\n"
  "v_add3_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_add3_u32 %0, %0, %1, %2\n"
  "v_add3_u32 %1, %1, %2, %3\n"
  "v_add3_u32 %2, %2, %3, %4\n"
  "v_add3_u32 %3, %3, %4, %5\n"
  "v_add3_u32 %4, %4, %5, %6\n"
  "v_add3_u32 %5, %5, %6, %7\n"
  "v_add3_u32 %6, %6, %7, %0\n"
  "v_add3_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_lshladd(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_lshl_add_u32 %0, %0, 4, %1\n"
  "v_lshl_add_u32 %1, %1, 4, %2\n"
  "v_lshl_add_u32 %2, %2, 4, %3\n"
  "v_lshl_add_u32 %3, %3, 4, %4\n"
  "v_lshl_add_u32 %4, %4, 4, %5\n"
  "v_lshl_add_u32 %5, %5, 4, %6\n"
  "v_lshl_add_u32 %6, %6, 4, %7\n"
  "v_lshl_add_u32 %7, %7, 4, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshl_add_u32 %0, %0, 4, %1\n"
  "v_lshl_add_u32 %1, %1, 4, %2\n"
  "v_lshl_add_u32 %2, %2, 4, %3\n"
  "v_lshl_add_u32 %3, %3, 4, %4\n"
  "v_lshl_add_u32 %4, %4, 4, %5\n"
  "v_lshl_add_u32 %5, %5, 4, %6\n"
  "v_lshl_add_u32 %6, %6, 4, %7\n"
  "v_lshl_add_u32 %7, %7, 4, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshl_add_u32 %0, %0, 4, %1\n"
  "v_lshl_add_u32 %1, %1, 4, %2\n"
  "v_lshl_add_u32 %2, %2, 4, %3\n"
  "v_lshl_add_u32 %3, %3, 4, %4\n"
  "v_lshl_add_u32 %4, %4, 4, %5\n"
  "v_lshl_add_u32 %5, %5, 4, %6\n"
  "v_lshl_add_u32 %6, %6, 4, %7\n"
  "v_lshl_add_u32 %7, %7, 4, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshl_add_u32 %0, %0, 4, %1\n"
  "v_lshl_add_u32 %1, %1, 4, %2\n"
  "v_lshl_add_u32 %2, %2, 4, %3\n"
  "v_lshl_add_u32 %3, %3, 4, %4\n"
  "v_lshl_add_u32 %4, %4, 4, %5\n"
  "v_lshl_add_u32 %5, %5, 4, %6\n"
  "v_lshl_add_u32 %6, %6, 4, %7\n"
  "v_lshl_add_u32 %7, %7, 4, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_bfi(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_bfi_b32 %0, %0, %1, %2\n"
  "v_bfi_b32 %1, %1, %2, %3\n"
  "v_bfi_b32 %2, %2, %3, %4\n"
  "v_bfi_b32 %3, %3, %4, %5\n"
  "v_bfi_b32 %4, %4, %5, %6\n"
  "v_bfi_b32 %5, %5, %6, %7\n"
  "v_bfi_b32 %6, %6, %7, %0\n"
  "v_bfi_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bfi_b32 %0, %0, %1, %2\n"
  "v_bfi_b32 %1, %1, %2, %3\n"
  "v_bfi_b32 %2, %2, %3, %4\n"
  "v_bfi_b32 %3, %3, %4, %5\n"
  "v_bfi_b32 %4, %4, %5, %6\n"
  "v_bfi_b32 %5, %5, %6, %7\n"
  "v_bfi_b32 %6, %6, %7, %0\n"
  "v_bfi_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bfi_b32 %0, %0, %1, %2\n"
  "v_bfi_b32 %1, %1, %2, %3\n"
  "v_bfi_b32 %2, %2, %3, %4\n"
  "v_bfi_b32 %3, %3, %4, %5\n"
  "v_bfi_b32 %4, %4, %5, %6\n"
  "v_bfi_b32 %5, %5, %6, %7\n"
  "v_bfi_b32 %6, %6, %7, %0\n"
  "v_bfi_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bfi_b32 %0, %0, %1, %2\n"
  "v_bfi_b32 %1, %1, %2, %3\n"
  "v_bfi_b32 %2, %2, %3, %4\n"
  "v_bfi_b32 %3, %3, %4, %5\n"
  "v_bfi_b32 %4, %4, %5, %6\n"
  "v_bfi_b32 %5, %5, %6, %7\n"
  "v_bfi_b32 %6, %6, %7, %0\n"
  "v_bfi_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_bitop3d(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8\n"
  "v_bitop3_b32 %1, %1, %2, %3 bitop3:0xe8\n"
  "v_bitop3_b32 %2, %2, %3, %4 bitop3:0xe8\n"
  "v_bitop3_b32 %3, %3, %4, %5 bitop3:0xe8\n"
  "v_bitop3_b32 %4, %4, %5, %6 bitop3:0xe8\n"
  "v_bitop3_b32 %5, %5, %6, %7 bitop3:0xe8\n"
  "v_bitop3_b32 %6, %6, %7, %0 bitop3:0xe8\n"
  "v_bitop3_b32 %7, %7, %0, %1 bitop3:0xe8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8\n"
  "v_bitop3_b32 %1, %1, %2, %3 bitop3:0xe8\n"
  "v_bitop3_b32 %2, %2, %3, %4 bitop3:0xe8\n"
  "v_bitop3_b32 %3, %3, %4, %5 bitop3:0xe8\n"
  "v_bitop3_b32 %4, %4, %5, %6 bitop3:0xe8\n"
  "v_bitop3_b32 %5, %5, %6, %7 bitop3:0xe8\n"
  "v_bitop3_b32 %6, %6, %7, %0 bitop3:0xe8\n"
  "v_bitop3_b32 %7, %7, %0, %1 bitop3:0xe8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8\n"
  "v_bitop3_b32 %1, %1, %2, %3 bitop3:0xe8\n"
  "v_bitop3_b32 %2, %2, %3, %4 bitop3:0xe8\n"
  "v_bitop3_b32 %3, %3, %4, %5 bitop3:0xe8\n"
  "v_bitop3_b32 %4, %4, %5, %6 bitop3:0xe8\n"
  "v_bitop3_b32 %5, %5, %6, %7 bitop3:0xe8\n"
  "v_bitop3_b32 %6, %6, %7, %0 bitop3:0xe8\n"
  "v_bitop3_b32 %7, %7, %0, %1 bitop3:0xe8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe8\n"
  "v_bitop3_b32 %1, %1, %2, %3 bitop3:0xe8\n"
  "v_bitop3_b32 %2, %2, %3, %4 bitop3:0xe8\n"
  "v_bitop3_b32 %3, %3, %4, %5 bitop3:0xe8\n"
  "v_bitop3_b32 %4, %4, %5, %6 bitop3:0xe8\n"
  "v_bitop3_b32 %5, %5, %6, %7 bitop3:0xe8\n"
  "v_bitop3_b32 %6, %6, %7, %0 bitop3:0xe8\n"
  "v_bitop3_b32 %7, %7, %0, %1 bitop3:0xe8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_bitop3c(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_bitop3_b32 %0, %0, 7, %1 bitop3:0xc8\n"
  "v_bitop3_b32 %1, %1, 7, %2 bitop3:0xc8\n"
  "v_bitop3_b32 %2, %2, 7, %3 bitop3:0xc8\n"
  "v_bitop3_b32 %3, %3, 7, %4 bitop3:0xc8\n"
  "v_bitop3_b32 %4, %4, 7, %5 bitop3:0xc8\n"
  "v_bitop3_b32 %5, %5, 7, %6 bitop3:0xc8\n"
  "v_bitop3_b32 %6, %6, 7, %7 bitop3:0xc8\n"
  "v_bitop3_b32 %7, %7, 7, %0 bitop3:0xc8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bitop3_b32 %0, %0, 7, %1 bitop3:0xc8\n"
  "v_bitop3_b32 %1, %1, 7, %2 bitop3:0xc8\n"
  "v_bitop3_b32 %2, %2, 7, %3 bitop3:0xc8\n"
  "v_bitop3_b32 %3, %3, 7, %4 bitop3:0xc8\n"
  "v_bitop3_b32 %4, %4, 7, %5 bitop3:0xc8\n"
  "v_bitop3_b32 %5, %5, 7, %6 bitop3:0xc8\n"
  "v_bitop3_b32 %6, %6, 7, %7 bitop3:0xc8\n"
  "v_bitop3_b32 %7, %7, 7, %0 bitop3:0xc8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bitop3_b32 %0, %0, 7, %1 bitop3:0xc8\n"
  "v_bitop3_b32 %1, %1, 7, %2 bitop3:0xc8\n"
  "v_bitop3_b32 %2, %2, 7, %3 bitop3:0xc8\n"
  "v_bitop3_b32 %3, %3, 7, %4 bitop3:0xc8\n"
  "v_bitop3_b32 %4, %4, 7, %5 bitop3:0xc8\n"
  "v_bitop3_b32 %5, %5, 7, %6 bitop3:0xc8\n"
  "v_bitop3_b32 %6, %6, 7, %7 bitop3:0xc8\n"
  "v_bitop3_b32 %7, %7, 7, %0 bitop3:0xc8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bitop3_b32 %0, %0, 7, %1 bitop3:0xc8\n"
  "v_bitop3_b32 %1, %1, 7, %2 bitop3:0xc8\n"
  "v_bitop3_b32 %2, %2, 7, %3 bitop3:0xc8\n"
  "v_bitop3_b32 %3, %3, 7, %4 bitop3:0xc8\n"
  "v_bitop3_b32 %4, %4, 7, %5 bitop3:0xc8\n"
  "v_bitop3_b32 %5, %5, 7, %6 bitop3:0xc8\n"
  "v_bitop3_b32 %6, %6, 7, %7 bitop3:0xc8\n"
  "v_bitop3_b32 %7, %7, 7, %0 bitop3:0xc8\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_or3d(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_or3_b32 %0, %0, %1, %2\n"
  "v_or3_b32 %1, %1, %2, %3\n"
  "v_or3_b32 %2, %2, %3, %4\n"
  "v_or3_b32 %3, %3, %4, %5\n"
  "v_or3_b32 %4, %4, %5, %6\n"
  "v_or3_b32 %5, %5, %6, %7\n"
  "v_or3_b32 %6, %6, %7, %0\n"
  "v_or3_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_or3_b32 %0, %0, %1, %2\n"
  "v_or3_b32 %1, %1, %2, %3\n"
  "v_or3_b32 %2, %2, %3, %4\n"
  "v_or3_b32 %3, %3, %4, %5\n"
  "v_or3_b32 %4, %4, %5, %6\n"
  "v_or3_b32 %5, %5, %6, %7\n"
  "v_or3_b32 %6, %6, %7, %0\n"
  "v_or3_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_or3_b32 %0, %0, %1, %2\n"
  "v_or3_b32 %1, %1, %2, %3\n"
  "v_or3_b32 %2, %2, %3, %4\n"
  "v_or3_b32 %3, %3, %4, %5\n"
  "v_or3_b32 %4, %4, %5, %6\n"
  "v_or3_b32 %5, %5, %6, %7\n"
  "v_or3_b32 %6, %6, %7, %0\n"
  "v_or3_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_or3_b32 %0, %0, %1, %2\n"
  "v_or3_b32 %1, %1, %2, %3\n"
  "v_or3_b32 %2, %2, %3, %4\n"
  "v_or3_b32 %3, %3, %4, %5\n"
  "v_or3_b32 %4, %4, %5, %6\n"
  "v_or3_b32 %5, %5, %6, %7\n"
  "v_or3_b32 %6, %6, %7, %0\n"
  "v_or3_b32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_lshlor2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_lshl_or_b32 %0, %0, 1, %1\n"
  "v_lshl_or_b32 %1, %1, 1, %2\n"
  "v_lshl_or_b32 %2, %2, 1, %3\n"
  "v_lshl_or_b32 %3, %3, 1, %4\n"
  "v_lshl_or_b32 %4, %4, 1, %5\n"
  "v_lshl_or_b32 %5, %5, 1, %6\n"
  "v_lshl_or_b32 %6, %6, 1, %7\n"
  "v_lshl_or_b32 %7, %7, 1, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshl_or_b32 %0, %0, 1, %1\n"
  "v_lshl_or_b32 %1, %1, 1, %2\n"
  "v_lshl_or_b32 %2, %2, 1, %3\n"
  "v_lshl_or_b32 %3, %3, 1, %4\n"
  "v_lshl_or_b32 %4, %4, 1, %5\n"
  "v_lshl_or_b32 %5, %5, 1, %6\n"
  "v_lshl_or_b32 %6, %6, 1, %7\n"
  "v_lshl_or_b32 %7, %7, 1, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshl_or_b32 %0, %0, 1, %1\n"
  "v_lshl_or_b32 %1, %1, 1, %2\n"
  "v_lshl_or_b32 %2, %2, 1, %3\n"
  "v_lshl_or_b32 %3, %3, 1, %4\n"
  "v_lshl_or_b32 %4, %4, 1, %5\n"
  "v_lshl_or_b32 %5, %5, 1, %6\n"
  "v_lshl_or_b32 %6, %6, 1, %7\n"
  "v_lshl_or_b32 %7, %7, 1, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshl_or_b32 %0, %0, 1, %1\n"
  "v_lshl_or_b32 %1, %1, 1, %2\n"
  "v_lshl_or_b32 %2, %2, 1, %3\n"
  "v_lshl_or_b32 %3, %3, 1, %4\n"
  "v_lshl_or_b32 %4, %4, 1, %5\n"
  "v_lshl_or_b32 %5, %5, 1, %6\n"
  "v_lshl_or_b32 %6, %6, 1, %7\n"
  "v_lshl_or_b32 %7, %7, 1, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_andor2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_and_or_b32 %0, %0, 63, %1\n"
  "v_and_or_b32 %1, %1, 63, %2\n"
  "v_and_or_b32 %2, %2, 63, %3\n"
  "v_and_or_b32 %3, %3, 63, %4\n"
  "v_and_or_b32 %4, %4, 63, %5\n"
  "v_and_or_b32 %5, %5, 63, %6\n"
  "v_and_or_b32 %6, %6, 63, %7\n"
  "v_and_or_b32 %7, %7, 63, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_or_b32 %0, %0, 63, %1\n"
  "v_and_or_b32 %1, %1, 63, %2\n"
  "v_and_or_b32 %2, %2, 63, %3\n"
  "v_and_or_b32 %3, %3, 63, %4\n"
  "v_and_or_b32 %4, %4, 63, %5\n"
  "v_and_or_b32 %5, %5, 63, %6\n"
  "v_and_or_b32 %6, %6, 63, %7\n"
  "v_and_or_b32 %7, %7, 63, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_or_b32 %0, %0, 63, %1\n"
  "v_and_or_b32 %1, %1, 63, %2\n"
  "v_and_or_b32 %2, %2, 63, %3\n"
  "v_and_or_b32 %3, %3, 63, %4\n"
  "v_and_or_b32 %4, %4, 63, %5\n"
  "v_and_or_b32 %5, %5, 63, %6\n"
  "v_and_or_b32 %6, %6, 63, %7\n"
  "v_and_or_b32 %7, %7, 63, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_or_b32 %0, %0, 63, %1\n"
  "v_and_or_b32 %1, %1, 63, %2\n"
  "v_and_or_b32 %2, %2, 63, %3\n"
  "v_and_or_b32 %3, %3, 63, %4\n"
  "v_and_or_b32 %4, %4, 63, %5\n"
  "v_and_or_b32 %5, %5, 63, %6\n"
  "v_and_or_b32 %6, %6, 63, %7\n"
  "v_and_or_b32 %7, %7, 63, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_bfev2(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_bfe_u32 %0, %0, %1, 1\n"
  "v_bfe_u32 %1, %1, %2, 1\n"
  "v_bfe_u32 %2, %2, %3, 1\n"
  "v_bfe_u32 %3, %3, %4, 1\n"
  "v_bfe_u32 %4, %4, %5, 1\n"
  "v_bfe_u32 %5, %5, %6, 1\n"
  "v_bfe_u32 %6, %6, %7, 1\n"
  "v_bfe_u32 %7, %7, %0, 1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bfe_u32 %0, %0, %1, 1\n"
  "v_bfe_u32 %1, %1, %2, 1\n"
  "v_bfe_u32 %2, %2, %3, 1\n"
  "v_bfe_u32 %3, %3, %4, 1\n"
  "v_bfe_u32 %4, %4, %5, 1\n"
  "v_bfe_u32 %5, %5, %6, 1\n"
  "v_bfe_u32 %6, %6, %7, 1\n"
  "v_bfe_u32 %7, %7, %0, 1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bfe_u32 %0, %0, %1, 1\n"
  "v_bfe_u32 %1, %1, %2, 1\n"
  "v_bfe_u32 %2, %2, %3, 1\n"
  "v_bfe_u32 %3, %3, %4, 1\n"
  "v_bfe_u32 %4, %4, %5, 1\n"
  "v_bfe_u32 %5, %5, %6, 1\n"
  "v_bfe_u32 %6, %6, %7, 1\n"
  "v_bfe_u32 %7, %7, %0, 1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_bfe_u32 %0, %0, %1, 1\n"
  "v_bfe_u32 %1, %1, %2, 1\n"
  "v_bfe_u32 %2, %2, %3, 1\n"
  "v_bfe_u32 %3, %3, %4, 1\n"
  "v_bfe_u32 %4, %4, %5, 1\n"
  "v_bfe_u32 %5, %5, %6, 1\n"
  "v_bfe_u32 %6, %6, %7, 1\n"
  "v_bfe_u32 %7, %7, %0, 1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_cndmask_s(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n"
  "v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n"
  "v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n"
  "v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n"
  "v_cndmask_b32_e64 %4, %4, %5, s[10:11]\n"
  "v_cndmask_b32_e64 %5, %5, %6, s[10:11]\n"
  "v_cndmask_b32_e64 %6, %6, %7, s[10:11]\n"
  "v_cndmask_b32_e64 %7, %7, %0, s[10:11]\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n"
  "v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n"
  "v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n"
  "v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n"
  "v_cndmask_b32_e64 %4, %4, %5, s[10:11]\n"
  "v_cndmask_b32_e64 %5, %5, %6, s[10:11]\n"
  "v_cndmask_b32_e64 %6, %6, %7, s[10:11]\n"
  "v_cndmask_b32_e64 %7, %7, %0, s[10:11]\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n"
  "v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n"
  "v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n"
  "v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n"
  "v_cndmask_b32_e64 %4, %4, %5, s[10:11]\n"
  "v_cndmask_b32_e64 %5, %5, %6, s[10:11]\n"
  "v_cndmask_b32_e64 %6, %6, %7, s[10:11]\n"
  "v_cndmask_b32_e64 %7, %7, %0, s[10:11]\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n"
  "v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n"
  "v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n"
  "v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n"
  "v_cndmask_b32_e64 %4, %4, %5, s[10:11]\n"
  "v_cndmask_b32_e64 %5, %5, %6, s[10:11]\n"
  "v_cndmask_b32_e64 %6, %6, %7, s[10:11]\n"
  "v_cndmask_b32_e64 %7, %7, %0, s[10:11]\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_xad(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_xad_u32 %0, %0, %1, %2\n"
  "v_xad_u32 %1, %1, %2, %3\n"
  "v_xad_u32 %2, %2, %3, %4\n"
  "v_xad_u32 %3, %3, %4, %5\n"
  "v_xad_u32 %4, %4, %5, %6\n"
  "v_xad_u32 %5, %5, %6, %7\n"
  "v_xad_u32 %6, %6, %7, %0\n"
  "v_xad_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_xad_u32 %0, %0, %1, %2\n"
  "v_xad_u32 %1, %1, %2, %3\n"
  "v_xad_u32 %2, %2, %3, %4\n"
  "v_xad_u32 %3, %3, %4, %5\n"
  "v_xad_u32 %4, %4, %5, %6\n"
  "v_xad_u32 %5, %5, %6, %7\n"
  "v_xad_u32 %6, %6, %7, %0\n"
  "v_xad_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_xad_u32 %0, %0, %1, %2\n"
  "v_xad_u32 %1, %1, %2, %3\n"
  "v_xad_u32 %2, %2, %3, %4\n"
  "v_xad_u32 %3, %3, %4, %5\n"
  "v_xad_u32 %4, %4, %5, %6\n"
  "v_xad_u32 %5, %5, %6, %7\n"
  "v_xad_u32 %6, %6, %7, %0\n"
  "v_xad_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_xad_u32 %0, %0, %1, %2\n"
  "v_xad_u32 %1, %1, %2, %3\n"
  "v_xad_u32 %2, %2, %3, %4\n"
  "v_xad_u32 %3, %3, %4, %5\n"
  "v_xad_u32 %4, %4, %5, %6\n"
  "v_xad_u32 %5, %5, %6, %7\n"
  "v_xad_u32 %6, %6, %7, %0\n"
  "v_xad_u32 %7, %7, %0, %1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_min(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_min_u32 %0, %0, %1\n"
  "v_min_u32 %1, %1, %2\n"
  "v_min_u32 %2, %2, %3\n"
  "v_min_u32 %3, %3, %4\n"
  "v_min_u32 %4, %4, %5\n"
  "v_min_u32 %5, %5, %6\n"
  "v_min_u32 %6, %6, %7\n"
  "v_min_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_min_u32 %0, %0, %1\n"
  "v_min_u32 %1, %1, %2\n"
  "v_min_u32 %2, %2, %3\n"
  "v_min_u32 %3, %3, %4\n"
  "v_min_u32 %4, %4, %5\n"
  "v_min_u32 %5, %5, %6\n"
  "v_min_u32 %6, %6, %7\n"
  "v_min_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_min_u32 %0, %0, %1\n"
  "v_min_u32 %1, %1, %2\n"
  "v_min_u32 %2, %2, %3\n"
  "v_min_u32 %3, %3, %4\n"
  "v_min_u32 %4, %4, %5\n"
  "v_min_u32 %5, %5, %6\n"
  "v_min_u32 %6, %6, %7\n"
  "v_min_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_min_u32 %0, %0, %1\n"
  "v_min_u32 %1, %1, %2\n"
  "v_min_u32 %2, %2, %3\n"
  "v_min_u32 %3, %3, %4\n"
  "v_min_u32 %4, %4, %5\n"
  "v_min_u32 %5, %5, %6\n"
  "v_min_u32 %6, %6, %7\n"
  "v_min_u32 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_pkadd(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_pk_add_u16 %0, %0, %1\n"
  "v_pk_add_u16 %1, %1, %2\n"
  "v_pk_add_u16 %2, %2, %3\n"
  "v_pk_add_u16 %3, %3, %4\n"
  "v_pk_add_u16 %4, %4, %5\n"
  "v_pk_add_u16 %5, %5, %6\n"
  "v_pk_add_u16 %6, %6, %7\n"
  "v_pk_add_u16 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_pk_add_u16 %0, %0, %1\n"
  "v_pk_add_u16 %1, %1, %2\n"
  "v_pk_add_u16 %2, %2, %3\n"
  "v_pk_add_u16 %3, %3, %4\n"
  "v_pk_add_u16 %4, %4, %5\n"
  "v_pk_add_u16 %5, %5, %6\n"
  "v_pk_add_u16 %6, %6, %7\n"
  "v_pk_add_u16 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_pk_add_u16 %0, %0, %1\n"
  "v_pk_add_u16 %1, %1, %2\n"
  "v_pk_add_u16 %2, %2, %3\n"
  "v_pk_add_u16 %3, %3, %4\n"
  "v_pk_add_u16 %4, %4, %5\n"
  "v_pk_add_u16 %5, %5, %6\n"
  "v_pk_add_u16 %6, %6, %7\n"
  "v_pk_add_u16 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_pk_add_u16 %0, %0, %1\n"
  "v_pk_add_u16 %1, %1, %2\n"
  "v_pk_add_u16 %2, %2, %3\n"
  "v_pk_add_u16 %3, %3, %4\n"
  "v_pk_add_u16 %4, %4, %5\n"
  "v_pk_add_u16 %5, %5, %6\n"
  "v_pk_add_u16 %6, %6, %7\n"
  "v_pk_add_u16 %7, %7, %0\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_sdwa(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
  "v_and_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ void __launch_bounds__(1024, 1) k_lshl_sdwa(u32* out, u32 seed) {
  u32 a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  for (int it = 0; it < ITER; ++it) {
    asm volatile(
  "v_lshlrev_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshlrev_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshlrev_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    asm volatile(
  "v_lshlrev_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
  "v_lshlrev_b32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

struct Case { const char* name; void (*fn)(u32*, u32); int per_iter; };
int main() {
  u32* out;
  hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<Case> cases = {
    {"v_add_u32", k_add, 32}, {"v_lshl_or_b32", k_lshl_or, 32}, {"v_and_or_b32", k_and_or, 32}, {"v_bfe_u32 const", k_bfe_const, 32},
    {"v_bfe_u32 var", k_bfe_var, 32}, {"v_alignbit_b32", k_alignbit, 32}, {"v_bitop3_b32", k_bitop3, 32}, {"v_perm_b32", k_perm, 32},
    {"v_bcnt_u32_b32", k_bcnt, 32}, {"v_mbcnt_lo", k_mbcnt, 32}, {"v_ffbl_b32", k_ffbl, 32}, {"v_or3_b32", k_or3, 32},
    {"v_cndmask_b32", k_cndmask, 32}, {"v_cmp_ne_u32", k_cmp, 32}, {"v_mov_dpp quad_perm", k_dpp_quad, 32}, {"v_mov_dpp row_ror:8", k_dpp_ror8, 32},
    {"v_mul_lo_u32", k_mul_lo, 32}, {"v_mad_u32_u24", k_mad24, 32},
    {"v_and_b32 v,v,v", k_and2, 32}, {"v_or_b32 v,v,v", k_or2, 32}, {"v_xor_b32 v,v,v", k_xor2, 32}, {"v_and_b32 v,63,v", k_andc, 32}, {"v_and_b32 v,0x1ff8,v", k_andlit, 32}, {"v_lshlrev_b32 v,2,v", k_lshlc, 32}, {"v_lshrrev_b32 v,13,v", k_lshrc, 32}, {"v_lshrrev_b32 v,v,v", k_lshrv, 32}, {"v_mov_b32 v,v", k_mov, 32}, {"v_add_u32 v,v,v", k_add2, 32}, {"v_sub_u32 v,v,v", k_sub2, 32}, {"v_add3_u32", k_add3, 32}, {"v_lshl_add_u32", k_lshladd, 32}, {"v_bfi_b32", k_bfi, 32}, {"v_bitop3_b32 3 regs", k_bitop3d, 32}, {"v_bitop3_b32 v,7,v", k_bitop3c, 32}, {"v_or3_b32 3 regs", k_or3d, 32}, {"v_lshl_or_b32 v,v,1,v", k_lshlor2, 32}, {"v_and_or_b32 v,v,63,v", k_andor2, 32}, {"v_bfe_u32 v,v,v,1", k_bfev2, 32}, {"v_cndmask_b32 e64 sgpr", k_cndmask_s, 32}, {"v_xad_u32", k_xad, 32}, {"v_min_u32", k_min, 32}, {"v_pk_add_u16", k_pkadd, 32}, {"v_and_b32_sdwa BYTE_1", k_sdwa, 32}, {"v_lshlrev_b32_sdwa BYTE_1", k_lshl_sdwa, 32},
    {"v_lshlrev_b64 const", k_lshl64_const, 32}, {"v_lshrrev_b64 var", k_lshr64_var, 32}, {"v_lshl_add_u64", k_add64, 32},
    {"ds_read_u8 random (+lcg 2 valu)", k_ds_read_u8_random, 8}, {"ds_read_u8 lane=bank (+lcg)", k_ds_read_u8_samebank, 8},
    {"ds_swizzle_b32", k_ds_swizzle, 8}, {"ds_bpermute_b32", k_ds_bpermute, 8}, {"v_permlane32/16_swap", k_permlane32_swap, 8},
  };
  for (const Case& c : cases) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(c.fn, dim3(256), dim3(1024), 0, 0, out, (u32)rep);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // 16 waves per CU = 4 per SIMD; every SIMD issues ITER * per_iter * 4 wave instructions
    const double ninst = (double)ITER * c.per_iter * 4.0;
    printf("%-34s %8.3f ms  %6.2f ns per wave-instruction per SIMD  (= %5.2f cycles at 2.4 GHz)\n", c.name, ms, ms * 1e6 / ninst, ms * 1e6 / ninst * 2.4);
  }
  return 0;
}
