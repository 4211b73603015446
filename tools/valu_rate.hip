// VALU issue-rate microbenchmark for gfx950 (MI355X): cycles per wave64 instruction, per op class,
// at 1 / 2 / 3 / 4 / 8 waves per SIMD.  Settles whether the integer ops K1 is made of issue at 2 or
// at 4 cycles per wave (VERDICT r01 "weak #2").  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// Every kernel runs kIters x 32 copies of ONE instruction on 8 independent register chains (no
// dependent-issue stalls: a chain's next use is 8 instructions away), inside a workgroup of 256
// threads (one wave per SIMD); `wps` workgroups per CU are forced with dynamic LDS and the grid is
// 256 CUs x wps, so each SIMD holds exactly `wps` waves.  Reported: shader cycles per instruction per
// SIMD = (s_memtime delta of the slowest wave) / (instructions issued on that SIMD), and the same from
// wall time at the clock the run sustained.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 2048;
constexpr int kUnroll = 32;     // instructions per iteration

// One instruction on chain register R (in/out), with two loop-invariant VGPR operands A, B.
#define DEF_OP(NAME, ASM)                                                                         \
  struct NAME {                                                                                   \
    static constexpr const char* name = #NAME;                                                    \
    static __device__ __forceinline__ void run(uint32_t& r, uint32_t a, uint32_t b) {             \
      asm volatile(ASM : "+v"(r) : "v"(a), "v"(b));                                               \
    }                                                                                             \
  };
// the same for instructions that write vcc / SGPRs / an AGPR (the clobber list costs an s_nop per statement)
#define DEF_OPC(NAME, ASM)                                                                        \
  struct NAME {                                                                                   \
    static constexpr const char* name = #NAME;                                                    \
    static __device__ __forceinline__ void run(uint32_t& r, uint32_t a, uint32_t b) {             \
      asm volatile(ASM : "+v"(r) : "v"(a), "v"(b)); \
    }                                                                                             \
  };
// 64-bit chain register
#define DEF_OP64(NAME, ASM)                                                                       \
  struct NAME {                                                                                   \
    static constexpr const char* name = #NAME;                                                    \
    static __device__ __forceinline__ void run(uint64_t& r, uint32_t a, uint32_t b) {             \
      asm volatile(ASM : "+v"(r) : "v"(a), "v"(b));                                               \
    }                                                                                             \
    typedef uint64_t is64;                                                                        \
  };

DEF_OP(v_mov_b32, "v_mov_b32 %0, %1")
DEF_OP(v_add_u32, "v_add_u32 %0, %0, %1")
DEF_OP(v_and_b32, "v_and_b32 %0, %0, %1")
DEF_OP(v_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
DEF_OP(v_lshrrev_b32_v, "v_lshrrev_b32 %0, %1, %0")
DEF_OP(v_add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEF_OP(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
DEF_OP(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1")
DEF_OP(v_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
DEF_OP(v_or3_b32, "v_or3_b32 %0, %0, %1, %2")
DEF_OP(v_xad_u32, "v_xad_u32 %0, %0, %1, %2")
DEF_OP(v_bfe_u32, "v_bfe_u32 %0, %0, 3, 7")
DEF_OP(v_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
DEF_OP(v_perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEF_OP(v_alignbit_b32, "v_alignbit_b32 %0, %0, %1, %2")
DEF_OP(v_alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, %2")
DEF_OP(v_ffbh_u32, "v_ffbh_u32 %0, %0")
DEF_OP(v_ffbl_b32, "v_ffbl_b32 %0, %0")
DEF_OP(v_bcnt_u32_b32, "v_bcnt_u32_b32 %0, %0, %1")
DEF_OP(v_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %0, %1")
DEF_OP(v_min_u32, "v_min_u32 %0, %0, %1")
DEF_OP(v_med3_i32, "v_med3_i32 %0, %0, %1, %2")
DEF_OP(v_min3_u32, "v_min3_u32 %0, %0, %1, %2")
DEF_OP(v_sad_u8, "v_sad_u8 %0, %0, %1, %2")
DEF_OPC(v_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF_OPC(v_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_OP(v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEF_OP(v_mul_i32_i24, "v_mul_i32_i24 %0, %0, %1")
DEF_OP(v_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEF_OP(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF_OP(v_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEF_OP(v_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
DEF_OP(v_mad_u32_u16_opsel, "v_mad_u32_u16 %0, %0, %1, %2 op_sel:[1,1,0,0]")
DEF_OP(v_mad_i32_i16, "v_mad_i32_i16 %0, %0, %1, %2")
DEF_OP(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_OP(v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_OP(v_dot2_i32_i16, "v_dot2_i32_i16 %0, %0, %1, %2")
DEF_OP(v_dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
DEF_OP(v_dot4_i32_i8, "v_dot4_i32_i8 %0, %0, %1, %2")
DEF_OP(v_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
DEF_OP(v_dot8_u32_u4, "v_dot8_u32_u4 %0, %0, %1, %2")
DEF_OP(v_pk_add_i16, "v_pk_add_i16 %0, %0, %1")
DEF_OP(v_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEF_OP(v_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
DEF_OP(v_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 3, %0")
DEF_OP(v_pk_ashrrev_i16, "v_pk_ashrrev_i16 %0, 2, %0")
DEF_OP(v_pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, %1, %0")
DEF_OP(v_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF_OP(v_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
DEF_OP(v_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEF_OP(v_pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
DEF_OP(v_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
DEF_OP(v_mad_u16, "v_mad_u16 %0, %0, %1, %2")
DEF_OP(v_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
DEF_OP(v_add_u16, "v_add_u16 %0, %0, %1")
DEF_OP(v_add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_1")
DEF_OP(v_mul_u32_u24_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_0")
DEF_OP(v_mov_b32_sdwa, "v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0")
DEF_OP(v_lshrrev_b32_sdwa, "v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD")
DEF_OP(v_mov_b32_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_OP(v_add_u32_dpp, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_OP(v_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_OP(v_add_f32, "v_add_f32 %0, %0, %1")
DEF_OP(v_mul_f32, "v_mul_f32 %0, %0, %1")
DEF_OP(v_fmac_f32, "v_fmac_f32 %0, %1, %2")
DEF_OP(v_floor_f32, "v_floor_f32 %0, %0")
DEF_OP(v_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
DEF_OP(v_cvt_f32_ubyte2, "v_cvt_f32_ubyte2 %0, %0")
DEF_OP(v_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
DEF_OP(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
DEF_OP(v_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
DEF_OP(v_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
DEF_OP(v_cvt_pk_u16_u32, "v_cvt_pk_u16_u32 %0, %0, %1")
DEF_OP(v_cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1")
DEF_OP(v_cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %0, %1, %2")
DEF_OP(v_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEF_OP(v_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
DEF_OP(v_fma_f16, "v_fma_f16 %0, %0, %1, %2")
DEF_OP(v_fma_mix_f32, "v_fma_mix_f32 %0, %0, %1, %2")
DEF_OP(v_frexp_exp_i32_f32, "v_frexp_exp_i32_f32 %0, %0")
DEF_OP(ds_bpermute_b32, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
DEF_OP(ds_swizzle_b32, "ds_swizzle_b32 %0, %0 offset:0x041f\n s_waitcnt lgkmcnt(0)")

DEF_OP(v_or_b32, "v_or_b32 %0, %0, %1")
DEF_OP(v_xor_b32, "v_xor_b32 %0, %0, %1")
DEF_OP(v_sub_u32, "v_sub_u32 %0, %0, %1")
DEF_OP(v_subrev_u32, "v_subrev_u32 %0, %0, %1")
DEF_OP(v_not_b32, "v_not_b32 %0, %0")
DEF_OP(v_lshlrev_b32_v, "v_lshlrev_b32 %0, %1, %0")
DEF_OP(v_lshrrev_b32_c, "v_lshrrev_b32 %0, 3, %0")
DEF_OP(v_ashrrev_i32_c, "v_ashrrev_i32 %0, 3, %0")
DEF_OP(v_ashrrev_i32_v, "v_ashrrev_i32 %0, %1, %0")
DEF_OP(v_max_i32, "v_max_i32 %0, %0, %1")
DEF_OP(v_max_u32, "v_max_u32 %0, %0, %1")
DEF_OP(v_add_u32_c, "v_add_u32 %0, 7, %0")
DEF_OP(v_and_b32_lit, "v_and_b32 %0, 0x7fff7fff, %0")
DEF_OP(v_and_b32_s, "v_and_b32 %0, s0, %0")
DEF_OP(v_mov_b32_c, "v_mov_b32 %0, 0")
DEF_OPC(v_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
DEF_OPC(v_addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF_OPC(v_cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1")
DEF_OPC(v_cmp_lt_u32_s, "v_cmp_lt_u32 s[20:21], %0, %1")
DEF_OPC(v_cndmask_b32_s, "v_cndmask_b32 %0, %0, %1, s[22:23]")
DEF_OP(v_max_f32, "v_max_f32 %0, %0, %1")
DEF_OP(v_sub_f32, "v_sub_f32 %0, %0, %1")
DEF_OP(v_mac_like_fma_c, "v_fma_f32 %0, %0, 2.0, %1")
DEF_OP(v_cvt_f16_f32, "v_cvt_f16_f32 %0, %0")
DEF_OP(v_sub_u16, "v_sub_u16 %0, %0, %1")
DEF_OP(v_lshlrev_b16, "v_lshlrev_b16 %0, 3, %0")
DEF_OP(v_max_i16, "v_max_i16 %0, %0, %1")
DEF_OPC(v_accvgpr_wr_rd, "v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0")
DEF_OP(v_bfrev_b32, "v_bfrev_b32 %0, %0")
DEF_OP(v_sat_pk_u8_i16, "v_sat_pk_u8_i16 %0, %0")
DEF_OP(v_cvt_pk_u8_like_perm2, "v_perm_b32 %0, %0, %1, s0")
DEF_OP(mix_add_perm, "v_add_u32 %0, %0, %1\n v_perm_b32 %0, %0, %1, %2")
DEF_OP(mix_add_add_perm, "v_add_u32 %0, %0, %1\n v_and_b32 %0, %0, %2\n v_perm_b32 %0, %0, %1, %2")
DEF_OP(mix_fma_perm, "v_fma_f32 %0, %0, %1, %2\n v_perm_b32 %0, %0, %1, %2")
DEF_OP(mix_mov_dot2, "v_mov_b32 %0, %1\n v_dot2_i32_i16 %0, %0, %1, %2")
DEF_OPC(mix_salu_perm, "s_add_u32 s24, s24, 1\n v_perm_b32 %0, %0, %1, %2")
DEF_OPC(mix_salu2_add, "s_add_u32 s24, s24, 1\n s_lshl_b32 s25, s24, 1\n v_add_u32 %0, %0, %1")
DEF_OP(ds_read_b32_op, "ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)")
DEF_OP64(v_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
DEF_OP64(v_lshlrev_b64_v, "v_lshlrev_b64 %0, %1, %0")
DEF_OP64(v_lshrrev_b64_v, "v_lshrrev_b64 %0, %1, %0")
DEF_OP64(v_mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
DEF_OP64(v_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
DEF_OP64(v_pk_add_f32, "v_pk_add_f32 %0, %0, %0")
DEF_OP64(v_pk_mul_f32, "v_pk_mul_f32 %0, %0, %0")
DEF_OP64(v_pk_mov_b32, "v_pk_mov_b32 %0, %0, %0")
DEF_OP64(v_add_f64, "v_add_f64 %0, %0, %0")


// ---- two independent streams: %0 = chain register of stream 1, %3 = chain register of stream 2
#define DEF_OP2(NAME, ASM)                                                                        \
  struct NAME {                                                                                   \
    static constexpr const char* name = #NAME;                                                    \
    static __device__ __forceinline__ void run2(uint32_t& r, uint32_t& q, uint32_t a, uint32_t b) { \
      asm volatile(ASM : "+v"(r), "+v"(q) : "v"(a), "v"(b));                                      \
    }                                                                                             \
  };
DEF_OP2(ind_add_perm, "v_add_u32 %0, %0, %2\n v_perm_b32 %1, %1, %2, %3")
DEF_OP2(ind_add_add_perm_perm, "v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %1, %1, %3, %2")
DEF_OP2(ind_add_and, "v_add_u32 %0, %0, %2\n v_and_b32 %1, %1, %3")
DEF_OP2(ind_fma_perm, "v_fma_f32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3")
DEF_OP2(ind_fma_add, "v_fma_f32 %0, %0, %2, %3\n v_add_u32 %1, %1, %2")
DEF_OP2(ind_perm_dot2, "v_perm_b32 %0, %0, %2, %3\n v_dot2_i32_i16 %1, %1, %2, %3")
DEF_OP2(ind_add_x3_perm, "v_add_u32 %0, %0, %2\n v_and_b32 %1, %1, %3\n v_xor_b32 %0, %0, %3\n v_perm_b32 %1, %1, %2, %3")
DEF_OP2(dep_add_add, "v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3")
DEF_OP2(dep_perm_perm, "v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %0, %0, %3, %2")
DEF_OP2(ind_add_cmp, "v_add_u32 %0, %0, %2\n v_cmp_lt_u32 vcc, %1, %2")
DEF_OP2(ind_add_dsread, "v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n ds_read_b32 %1, %2\n s_waitcnt lgkmcnt(0)")

template <typename OP>
__global__ __launch_bounds__(256) void rate_kernel2(uint32_t a, uint32_t b, unsigned long long* cycles, uint32_t* sink) {
  extern __shared__ unsigned char lds[];
  uint32_t r[4], q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = threadIdx.x * 2654435761u + i * 40503u + a; q[i] = r[i] ^ 0x5555u; }
  __syncthreads();
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) OP::run2(r[u & 3], q[u & 3], a & 0xfc, b);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) x ^= r[i] ^ q[i];
  if (x == 0x12345u && lds[a & 1023] == 77) sink[0] = 1;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* c = cycles + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    c[0] = t1 - t0; c[1] = w0; c[2] = w1; c[3] = 0;
  }
}

template <typename T> struct chain_type { typedef uint32_t type; };
template <typename T> struct has64 { template <typename U> static char t(typename U::is64*); template <typename U> static long t(...); static constexpr bool v = sizeof(t<T>(nullptr)) == 1; };

template <typename OP, typename R>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t a, uint32_t b, unsigned long long* cycles, uint32_t* sink) {
  extern __shared__ unsigned char lds[];
  R r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = static_cast<R>(threadIdx.x * 2654435761u + i * 40503u + a);
  __syncthreads();
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) OP::run(r[u & 7], a, b);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  R x = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) x ^= r[i];
  if (static_cast<uint32_t>(x) == 0x12345u && lds[a & 1023] == 77) sink[0] = 1;   // keep everything alive
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* c = cycles + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    c[0] = t1 - t0; c[1] = w0; c[2] = w1; c[3] = 0;
  }
}

struct Result { std::string name; double cyc[5]; double wall[5]; double ghz[5]; double overlap[5]; };

template <typename OP, typename K>
Result run_kernel(K kern, unsigned long long* d_cycles, uint32_t* d_sink, int n_cu, size_t max_lds) {
  Result res; res.name = OP::name;
  const int wps_list[5] = {1, 2, 3, 4, 8};
  for (int wi = 0; wi < 5; ++wi) {
    const int wps = wps_list[wi];
    const int grid = n_cu * wps;
    // dynamic LDS so that exactly `wps` workgroups fit a CU (160 KiB; 64 KiB cap per workgroup)
    size_t lds = max_lds / wps - 2048; lds &= ~size_t(1023);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(max_lds)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<grid, 256, lds>>>(3u, 5u, d_cycles, d_sink);     // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    kern<<<grid, 256, lds>>>(3u, 5u, d_cycles, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * 16);
    CHECK(hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0, wavg = 0; unsigned long long wmin = ~0ull, wmax = 0;
    for (int i = 0; i < grid * 4; ++i) {
      avg += double(h[4 * i]); wavg += double(h[4 * i + 2] - h[4 * i + 1]);
      if (h[4 * i + 1] < wmin) wmin = h[4 * i + 1];
      if (h[4 * i + 2] > wmax) wmax = h[4 * i + 2];
    }
    avg /= grid * 4; wavg /= grid * 4;
    res.ghz[wi] = avg / (wavg * 10.0);                 // wall_clock64 ticks at 100 MHz = 10 ns
    res.overlap[wi] = ms * 1e-3 * 2.4e9 / (double(kIters) * kUnroll * wps);   // host events, cycles at 2.4 GHz      // 1.0 = every wave ran for the whole kernel (all co-resident)
    const double instr_per_wave = double(kIters) * kUnroll;
    // a SIMD holds `wps` waves that all issue instr_per_wave instructions during ~avg cycles
    res.cyc[wi] = avg / (instr_per_wave * wps);
    res.wall[wi] = double(wmax - wmin) * 10e-9 * 2.4e9 / (instr_per_wave * wps);   // device wall clock, cycles at the nominal 2.4 GHz
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  }
  return res;
}

static void print_result(const Result& r) {
  printf("%-22s", r.name.c_str());
  for (int i = 0; i < 5; ++i) printf(" %6.2f", r.cyc[i]);
  printf("   |");
  for (int i = 0; i < 5; ++i) printf(" %6.2f", r.wall[i]);
  printf("   | GHz %4.2f %4.2f  ev", r.ghz[0], r.ghz[4]);
  for (int i = 0; i < 5; ++i) printf(" %5.2f", r.overlap[i]);
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  const int n_cu = prop.multiProcessorCount;
  unsigned long long* d_cycles; uint32_t* d_sink;
  CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * n_cu * 8 * 16));
  CHECK(hipMalloc(&d_sink, 64));
  std::vector<Result> rs;
  printf("# %s, %d CUs; s_memtime ticks per wave64 instruction per SIMD (mean over waves) | device wall clock, cycles at 2.4 GHz | s_memtime GHz at 1 and 8 waves/SIMD | fraction of the kernel a wave was running (3, 8 waves/SIMD)\n", prop.name, n_cu);
  printf("# waves/SIMD:            1      2      3      4      8   |  wall: 1      2      3      4      8\n");
  const size_t clk = prop.sharedMemPerBlock;   // 160 KiB on gfx950: one workgroup can take the whole CU
  printf("# sharedMemPerBlock %zu\n", clk);
#define RUN(OP) fprintf(stderr, "%s\n", OP::name); rs.push_back(run_kernel<OP>(rate_kernel<OP, uint32_t>, d_cycles, d_sink, n_cu, clk)); print_result(rs.back());
#define RUN64(OP) fprintf(stderr, "%s\n", OP::name); rs.push_back(run_kernel<OP>(rate_kernel<OP, uint64_t>, d_cycles, d_sink, n_cu, clk));
#define RUN2(OP) fprintf(stderr, "%s\n", OP::name); rs.push_back(run_kernel<OP>(rate_kernel2<OP>, d_cycles, d_sink, n_cu, clk)); print_result(rs.back()); print_result(rs.back());
  RUN(v_mov_b32) RUN(v_add_u32) RUN(v_and_b32) RUN(v_lshlrev_b32) RUN(v_lshrrev_b32_v) RUN(v_add3_u32)
  RUN(v_lshl_add_u32) RUN(v_lshl_or_b32) RUN(v_and_or_b32) RUN(v_or3_b32) RUN(v_xad_u32) RUN(v_bfe_u32) RUN(v_bfi_b32)
  RUN(v_perm_b32) RUN(v_alignbit_b32) RUN(v_alignbyte_b32) RUN(v_ffbh_u32) RUN(v_ffbl_b32) RUN(v_bcnt_u32_b32)
  RUN(v_mbcnt_lo) RUN(v_min_u32) RUN(v_med3_i32) RUN(v_min3_u32) RUN(v_sad_u8) RUN(v_cmp_cndmask) RUN(v_cndmask_b32)
  RUN(v_mul_u32_u24) RUN(v_mul_i32_i24) RUN(v_mul_hi_u32_u24) RUN(v_mad_u32_u24) RUN(v_mad_i32_i24)
  RUN(v_mad_u32_u16) RUN(v_mad_u32_u16_opsel) RUN(v_mad_i32_i16) RUN(v_mul_lo_u32) RUN(v_mul_hi_u32)
  RUN(v_dot2_i32_i16) RUN(v_dot2_u32_u16) RUN(v_dot4_i32_i8) RUN(v_dot4_u32_u8) RUN(v_dot8_u32_u4)
  RUN(v_pk_add_i16) RUN(v_pk_add_u16) RUN(v_pk_sub_i16) RUN(v_pk_lshlrev_b16) RUN(v_pk_ashrrev_i16) RUN(v_pk_lshrrev_b16)
  RUN(v_pk_max_i16) RUN(v_pk_min_u16) RUN(v_pk_mul_lo_u16) RUN(v_pk_mad_i16) RUN(v_pk_mad_u16)
  RUN(v_mad_u16) RUN(v_mul_lo_u16) RUN(v_add_u16)
  RUN(v_add_u32_sdwa) RUN(v_mul_u32_u24_sdwa) RUN(v_mov_b32_sdwa) RUN(v_lshrrev_b32_sdwa)
  RUN(v_mov_b32_dpp) RUN(v_add_u32_dpp)
  RUN(v_fma_f32) RUN(v_add_f32) RUN(v_mul_f32) RUN(v_fmac_f32) RUN(v_floor_f32)
  RUN(v_cvt_f32_ubyte0) RUN(v_cvt_f32_ubyte2) RUN(v_cvt_f32_i32) RUN(v_cvt_f32_u32) RUN(v_cvt_i32_f32) RUN(v_cvt_u32_f32)
  RUN(v_cvt_pk_u16_u32) RUN(v_cvt_pk_i16_i32) RUN(v_cvt_pk_u8_f32)
  RUN(v_pk_fma_f16) RUN(v_pk_mul_f16) RUN(v_fma_f16) RUN(v_fma_mix_f32) RUN(v_frexp_exp_i32_f32)
  RUN(ds_bpermute_b32) RUN(ds_swizzle_b32)
  RUN(v_or_b32) RUN(v_xor_b32) RUN(v_sub_u32) RUN(v_subrev_u32) RUN(v_not_b32) RUN(v_lshlrev_b32_v) RUN(v_lshrrev_b32_c) RUN(v_ashrrev_i32_c) RUN(v_ashrrev_i32_v) RUN(v_max_i32) RUN(v_max_u32) RUN(v_add_u32_c) RUN(v_and_b32_lit) RUN(v_and_b32_s) RUN(v_mov_b32_c) RUN(v_add_co_u32) RUN(v_addc_co_u32) RUN(v_cmp_lt_u32) RUN(v_cmp_lt_u32_s) RUN(v_cndmask_b32_s) RUN(v_max_f32) RUN(v_sub_f32) RUN(v_mac_like_fma_c) RUN(v_cvt_f16_f32) RUN(v_sub_u16) RUN(v_lshlrev_b16) RUN(v_max_i16) RUN(v_accvgpr_wr_rd) RUN(v_bfrev_b32) RUN(v_sat_pk_u8_i16) RUN(v_cvt_pk_u8_like_perm2) RUN(mix_add_perm) RUN(mix_add_add_perm) RUN(mix_fma_perm) RUN(mix_mov_dot2)
  printf("# two-stream rows: cycles per STATEMENT (2, 4, 2, 2, 2, 2, 4, 2, 2, 2, 6 instructions)\n");
  RUN2(ind_add_perm) RUN2(ind_add_add_perm_perm) RUN2(ind_add_and) RUN2(ind_fma_perm) RUN2(ind_fma_add) RUN2(ind_perm_dot2) RUN2(ind_add_x3_perm) RUN2(dep_add_add) RUN2(dep_perm_perm) RUN2(ind_add_cmp) RUN2(ind_add_dsread)
  RUN64(v_lshlrev_b64) RUN64(v_lshlrev_b64_v) RUN64(v_lshrrev_b64_v) RUN64(v_mad_u64_u32)
  RUN64(v_pk_fma_f32) RUN64(v_pk_add_f32) RUN64(v_pk_mul_f32) RUN64(v_pk_mov_b32) RUN64(v_add_f64)
  return 0;
}
