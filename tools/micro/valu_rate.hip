// Issue rate of the vector instructions the fused kernels are made of, per SIMD of a gfx950 compute unit, with 1, 2 and
// 4 waves resident per SIMD and 1, 2 or 8 independent register chains per wave.  One workgroup of 256 x W threads per
// compute unit (waves w and w + 4 share a SIMD, tools/micro/simd_map.hip); every wave runs ITER iterations of 64
// instructions of one kind written in inline assembly (nothing for the compiler to fold, fuse or pack).  Reported:
// quad-cycles (4 clocks at the device's peak clock) per wave-instruction and SIMD -- 1.00 means "one instruction per
// quad-cycle and SIMD", 0.50 means two waves' instructions overlap.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

#define ITER 4000
#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define R64(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S)

enum Kind {
  ADD_F32, FMA_F32, PK_ADD_F32, PK_FMA_F32, ADD_F64, MUL_F64, FMA_F64, CVT_F64_F32, CVT_F32_F64, LDEXP_F64, MOV_B32,
  MOV_DPP, ADD_SAMEBANK, ADD_DIFFBANK, ADD_SRC_DST_SAME_BANK, ADD_LONG_LOOP, DPP_WAVE_SHR, DPP_WAVE_SHL, DPP_ROW_BCAST15, DPP_ADD_WAVE_SHR, DS_BPERMUTE, DS_SWIZZLE, CNDMASK, CND_VCC_SET, CND_E64_SGPR, CND_E64_OTHER_DST, BFI_B32, AND_B32, ADD_U32, CMP_VCC, CMP_SGPR, CND_E64_VCC, MIX_F64_F32, MIX_F64_MOV, MIX_CVT_ADD64, MIX_CND32_ADD64, MIX_CND32_3ADD64, MIX7_DPP, MIX7_CVT, MIX7_MULF64, MIX7_SALU, MIX7_CND64, MIX7_MOV, MIX7_NOP, MIX7_DPP_ROW, MIX56_DPP8, MIX7_SWIZZLE, MIX7_BPERM, MIX7_READLANE, MIX7_PERMLANE, MIX248_DPP8, MIX1016_DPP8, MIX1016_DPP8_STAGGER, MIX7_DSREAD, MIX7_DSWRITE, MIX7_EXEC, MIX7_BUFLOAD, MIX7_WAITCNT, MIX63_BARRIER, MIX7_CVT32, MIX7_MUL_F32, MIX7_FMA_F32, NKINDS
};
static const char* kind_name[NKINDS] = {
    "v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_f32",
    "v_cvt_f32_f64", "v_ldexp_f64", "v_mov_b32", "v_mov_b32 dpp row_shr:1", "v_add_f32 v10,v14,v18 (one bank)", "v_add_f32 v10,v15,v17 (3 banks)", "v_add_f32 v10,v14,v17 (dst=src0 bank)", "v_add_f32, 40 KB loop body", "v_mov_b32 dpp wave_shr:1", "v_mov_b32 dpp wave_shl:1", "v_mov_b32 dpp row_bcast:15", "v_add_f32 dpp wave_shr:1", "ds_bpermute_b32 (+waitcnt per 8)", "ds_swizzle_b32 (+waitcnt per 8)", "v_cndmask_b32 vcc (never written)", "v_cndmask_b32 vcc (v_cmp before)", "v_cndmask_b32_e64 sgpr pair", "v_cndmask_b32_e64 dst != src", "v_bfi_b32", "v_and_b32", "v_add_u32", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32_e64 sgpr pair", "v_cndmask_b32_e64 vcc", "add_f64,add_f32 alternating",
    "add_f64,mov_b32 alternating", "cvt_f64_f32,add_f64 alternating", "add_f64,cndmask_e32 vcc alternating", "3 add_f64,cndmask_e32 vcc (per 4)", "7 add_f32 + dpp wave_shr", "7 add_f32 + cvt_f64_f32", "7 add_f32 + mul_f64", "7 add_f32 + s_and_b64", "7 add_f32 + cndmask_e64", "7 add_f32 + v_mov_b32", "7 add_f32 + s_nop 0", "7 add_f32 + dpp row_shr:1", "56 add_f32 + 8 dpp wave_shr in a row", "7 add_f32 + ds_swizzle", "7 add_f32 + ds_bpermute", "7 add_f32 + v_readlane", "7 add_f32 + v_permlane32_swap", "248 add_f32 + 8 dpp in a row", "1016 add_f32 + 8 dpp in a row", "1016 add_f32 + 8 dpp, odd waves shifted by 512", "7 add_f32 + ds_read_b128", "7 add_f32 + ds_write_b32", "7 add_f32 + s_and_saveexec/s_mov exec", "7 add_f32 + global_load_dword", "7 add_f32 + s_waitcnt 0", "63 add_f32 + s_barrier", "7 add_f32 + v_cvt_f32_i32", "7 add_f32 + v_mul_f32", "7 add_f32 + v_fma_f32"};

template <int K, int NCH>
__global__ void __launch_bounds__(1024) rate(float* out, float seed) {
  float a[8];
  f2 p[8];
  double d[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + i + threadIdx.x;
    p[i] = f2{a[i], a[i] + 1.f};
    d[i] = a[i];
  }
  float x = seed * 0.5f;
  f2 x2 = f2{x, x};
  double xd = x, yd = 1.0 + 1e-9 * x;
  int two = 1;
  const unsigned lds_at = threadIdx.x * 16u;  // an LDS byte address per lane (the kernel has 100 KB of dynamic LDS)
  const float* gptr = out + (threadIdx.x & 63);
  for (int it = 0; it < ITER; ++it) {
// one asm statement per iteration: the compiler cannot put wait states between the 64 instructions (it does between
// dependent inline-asm statements, whatever they contain).  %0-%7: the chains, %8 / %9: loop-invariant inputs.
#define X8(I, a, b, c, d, e, f, g, h) I(a) I(b) I(c) I(d) I(e) I(f) I(g) I(h)
#define SEQ1(I) X8(I, 0, 0, 0, 0, 0, 0, 0, 0)
#define SEQ2(I) X8(I, 0, 1, 0, 1, 0, 1, 0, 1)
#define SEQ8(I) X8(I, 0, 1, 2, 3, 4, 5, 6, 7)
#define REP8(S) S S S S S S S S
#define BODY(I, T, IN0, IN1)                                                                                            \
  if constexpr (NCH == 1)                                                                                               \
    asm volatile(REP8(SEQ1(I)) : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]) : "v"(IN0), "v"(IN1) : "vcc", "s20", "s21"); \
  else if constexpr (NCH == 2)                                                                                          \
    asm volatile(REP8(SEQ2(I)) : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]) : "v"(IN0), "v"(IN1) : "vcc", "s20", "s21"); \
  else                                                                                                                  \
    asm volatile(REP8(SEQ8(I)) : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]) : "v"(IN0), "v"(IN1) : "vcc", "s20", "s21");
#define BODYM(I)                                                                                                        \
  if constexpr (NCH == 1)                                                                                               \
    asm volatile(SEQ1(I) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(x) : "vcc", "s20", "s21", "v20", "v21", "v22", "v23", "memory"); \
  else if constexpr (NCH == 2)                                                                                          \
    asm volatile(SEQ2(I) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(x) : "vcc", "s20", "s21", "v20", "v21", "v22", "v23", "memory"); \
  else                                                                                                                  \
    asm volatile(SEQ8(I) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(x) : "vcc", "s20", "s21", "v20", "v21", "v22", "v23", "memory");
#define BODYM2(I, IN1)                                                                                                       \
  if constexpr (NCH == 1)                                                                                               \
    asm volatile(SEQ1(I) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(IN1) : "vcc", "s20", "s21", "v20", "v21", "v22", "v23", "memory"); \
  else if constexpr (NCH == 2)                                                                                          \
    asm volatile(SEQ2(I) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(IN1) : "vcc", "s20", "s21", "v20", "v21", "v22", "v23", "memory"); \
  else                                                                                                                  \
    asm volatile(SEQ8(I) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(IN1) : "vcc", "s20", "s21", "v20", "v21", "v22", "v23", "memory");
#define BODYR(I) asm volatile(REP8(X8(I, 0, 0, 0, 0, 0, 0, 0, 0)) : : : "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19");
#define I_ADD_F32(k) "v_add_f32 %" #k ", %8, %" #k "\n\t"
#define I_FMA_F32(k) "v_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define I_PK_ADD(k) "v_pk_add_f32 %" #k ", %8, %" #k "\n\t"
#define I_PK_FMA(k) "v_pk_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define I_ADD_F64(k) "v_add_f64 %" #k ", %8, %" #k "\n\t"
#define I_MUL_F64(k) "v_mul_f64 %" #k ", %9, %" #k "\n\t"
#define I_FMA_F64(k) "v_fma_f64 %" #k ", %8, %9, %" #k "\n\t"
#define I_CVT_64_32(k) "v_cvt_f64_f32 %" #k ", %8\n\t"
#define I_CVT_32_64(k) "v_cvt_f32_f64 %" #k ", %8\n\t"
#define I_LDEXP(k) "v_ldexp_f64 %" #k ", %" #k ", %8\n\t"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define I_DPP(k) "v_mov_b32_dpp %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPP_WSHR(k) "v_mov_b32_dpp %" #k ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPP_WSHL(k) "v_mov_b32_dpp %" #k ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPP_BC15(k) "v_mov_b32_dpp %" #k ", %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define I_DPP_ADD(k) "v_add_f32_dpp %" #k ", %8, %" #k " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_BPERM(k) "ds_bpermute_b32 %" #k ", %9, %8\n\t"
#define I_SWZ(k) "ds_swizzle_b32 %" #k ", %8 offset:swizzle(BROADCAST,32,1)\n\t"
#define I_ADD_B1(k) "v_add_f32 v10, v14, v18\n\tv_add_f32 v11, v15, v19\n\t"
#define I_ADD_B3(k) "v_add_f32 v10, v15, v17\n\tv_add_f32 v11, v16, v18\n\t"
#define I_ADD_B2(k) "v_add_f32 v10, v14, v17\n\tv_add_f32 v11, v15, v18\n\t"
#define I_CND(k) "v_cndmask_b32 %" #k ", %8, %" #k ", vcc\n\t"
#define I_CND64(k) "v_cndmask_b32_e64 %" #k ", %8, %" #k ", s[20:21]\n\t"
#define I_CND64D(k) "v_cndmask_b32_e64 %" #k ", %8, %9, s[20:21]\n\t"
#define I_BFI(k) "v_bfi_b32 %" #k ", %9, %8, %" #k "\n\t"
#define I_AND(k) "v_and_b32 %" #k ", %8, %" #k "\n\t"
#define I_ADDU(k) "v_add_u32 %" #k ", %8, %" #k "\n\t"
#define I_CMPV(k) "v_cmp_lt_f32 vcc, %8, %" #k "\n\t"
#define I_CMPS(k) "v_cmp_lt_f32_e64 s[20:21], %8, %" #k "\n\t"
#define I_CND64V(k) "v_cndmask_b32_e64 %" #k ", %8, %" #k ", vcc\n\t"
#define I_MIX4(k) "v_add_f64 %" #k ", %8, %" #k "\n\tv_cndmask_b32_e32 %9, %9, %9, vcc\n\t"
#define I_MIX5(k) "v_add_f64 %" #k ", %8, %" #k "\n\tv_add_f64 %" #k ", %8, %" #k "\n\tv_add_f64 %" #k ", %8, %" #k "\n\tv_cndmask_b32_e32 %9, %9, %9, vcc\n\t"
#define A7(k) "v_add_f32 %" #k ", %8, %" #k "\n\tv_add_f32 %" #k ", %8, %" #k "\n\tv_add_f32 %" #k ", %8, %" #k "\n\tv_add_f32 %" #k ", %8, %" #k "\n\tv_add_f32 %" #k ", %8, %" #k "\n\tv_add_f32 %" #k ", %8, %" #k "\n\tv_add_f32 %" #k ", %8, %" #k "\n\t"
#define I_M7_DPP(k) A7(k) "v_mov_b32_dpp v20, v22 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_M7_CVT(k) A7(k) "v_cvt_f64_f32 v[20:21], v22\n\t"
#define I_M7_MUL(k) A7(k) "v_mul_f64 v[20:21], v[22:23], v[22:23]\n\t"
#define I_M7_SALU(k) A7(k) "s_mov_b32 s20, s21\n\t"
#define I_M7_CND(k) A7(k) "v_cndmask_b32_e64 v20, v22, v23, s[20:21]\n\t"
#define I_M7_MOV(k) A7(k) "v_mov_b32 v20, v22\n\t"
#define I_M7_NOP(k) A7(k) "s_nop 0\n\t"
#define I_M7_DPPR(k) A7(k) "v_mov_b32_dpp v20, v22 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_M7_SWZ(k) A7(k) "ds_swizzle_b32 v20, v22 offset:swizzle(BROADCAST,32,1)\n\t"
#define I_M7_BPERM(k) A7(k) "ds_bpermute_b32 v20, v22, v23\n\t"
#define I_M7_RDL(k) A7(k) "v_readlane_b32 s20, v22, 5\n\t"
#define I_M7_PERM(k) A7(k) "v_permlane32_swap_b32 v20, v22\n\t"
#define A1(k) "v_add_f32 %" #k ", %8, %" #k "\n\t"
#define D1(k) "v_mov_b32_dpp v20, v22 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_M7_DSR(k) A7(k) "ds_read_b128 v[20:23], %9\n\t"
#define I_M7_DSW(k) A7(k) "ds_write_b32 %9, %8\n\t"
#define I_M7_EXEC(k) A7(k) "s_and_saveexec_b64 s[20:21], vcc\n\ts_mov_b64 exec, s[20:21]\n\t"
#define I_M7_GLD(k) A7(k) "global_load_dword v20, %9, off\n\t"
#define I_M7_WAIT(k) A7(k) "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define I_M7_CVT32(k) A7(k) "v_cvt_f32_i32 v20, v22\n\t"
#define I_M7_MULF(k) A7(k) "v_mul_f32 v20, v22, v22\n\t"
#define I_M7_FMAF(k) A7(k) "v_fma_f32 v20, v22, v22, v22\n\t"
#define I_MIX1(k) "v_add_f64 %" #k ", %8, %" #k "\n\tv_add_f32 %9, %9, %9\n\t"
#define I_MIX2(k) "v_add_f64 %" #k ", %8, %" #k "\n\tv_mov_b32 %9, %9\n\t"
#define I_MIX3(k) "v_add_f64 %" #k ", %8, %" #k "\n\tv_cvt_f64_f32 %7, %9\n\t"
    if constexpr (K == ADD_F32) { BODY(I_ADD_F32, a, x, x) }
    if constexpr (K == FMA_F32) { BODY(I_FMA_F32, a, x, x) }
    if constexpr (K == PK_ADD_F32) { BODY(I_PK_ADD, p, x2, x2) }
    if constexpr (K == PK_FMA_F32) { BODY(I_PK_FMA, p, x2, x2) }
    if constexpr (K == ADD_F64) { BODY(I_ADD_F64, d, xd, yd) }
    if constexpr (K == MUL_F64) { BODY(I_MUL_F64, d, xd, yd) }
    if constexpr (K == FMA_F64) { BODY(I_FMA_F64, d, xd, yd) }
    if constexpr (K == CVT_F64_F32) { BODY(I_CVT_64_32, d, x, x) }
    if constexpr (K == CVT_F32_F64) { BODY(I_CVT_32_64, a, xd, xd) }
    if constexpr (K == LDEXP_F64) { BODY(I_LDEXP, d, two, two) }
    if constexpr (K == MOV_B32) { BODY(I_MOV, a, x, x) }
    if constexpr (K == MOV_DPP) { BODY(I_DPP, a, x, x) }
    if constexpr (K == ADD_SAMEBANK) { BODYR(I_ADD_B1) }
    if constexpr (K == ADD_DIFFBANK) { BODYR(I_ADD_B3) }
    if constexpr (K == ADD_SRC_DST_SAME_BANK) { BODYR(I_ADD_B2) }
    if constexpr (K == ADD_LONG_LOOP) { BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x)
                                       BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x)
                                       BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x)
                                       BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) BODY(I_ADD_F32, a, x, x) }
    if constexpr (K == DPP_WAVE_SHR) { BODY(I_DPP_WSHR, a, x, x) }
    if constexpr (K == DPP_WAVE_SHL) { BODY(I_DPP_WSHL, a, x, x) }
    if constexpr (K == DPP_ROW_BCAST15) { BODY(I_DPP_BC15, a, x, x) }
    if constexpr (K == DPP_ADD_WAVE_SHR) { BODY(I_DPP_ADD, a, x, x) }
    if constexpr (K == DS_BPERMUTE) { BODY(I_BPERM, a, x, two) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (K == DS_SWIZZLE) { BODY(I_SWZ, a, x, x) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (K == CNDMASK) { BODY(I_CND, a, x, x) }
    if constexpr (K == CND_VCC_SET) {
      if (it == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(a[0]) : "vcc");
      BODY(I_CND, a, x, x)
    }
    if constexpr (K == CND_E64_SGPR) { asm volatile("s_mov_b64 s[20:21], 0x0f0f33aa" : : : "s20", "s21"); BODY(I_CND64, a, x, x) }
    if constexpr (K == CND_E64_OTHER_DST) { asm volatile("s_mov_b64 s[20:21], 0x0f0f33aa" : : : "s20", "s21"); BODY(I_CND64D, a, x, x) }
    if constexpr (K == BFI_B32) { BODY(I_BFI, a, x, two) }
    if constexpr (K == AND_B32) { BODY(I_AND, a, x, x) }
    if constexpr (K == ADD_U32) { BODY(I_ADDU, a, x, x) }
    if constexpr (K == CMP_VCC) { BODY(I_CMPV, a, x, x) }
    if constexpr (K == CMP_SGPR) { BODY(I_CMPS, a, x, x) }
    if constexpr (K == CND_E64_VCC) { BODY(I_CND64V, a, x, x) }
    // pairs: 64 add_f64 + 64 others per iteration, counted as 128 instructions below
    if constexpr (K == MIX_F64_F32) { BODY(I_MIX1, d, xd, x) }
    if constexpr (K == MIX_F64_MOV) { BODY(I_MIX2, d, xd, x) }
    if constexpr (K == MIX_CVT_ADD64) { BODY(I_MIX3, d, xd, x) }
    if constexpr (K == MIX_CND32_ADD64) { BODY(I_MIX4, d, xd, x) }
    if constexpr (K == MIX_CND32_3ADD64) { BODY(I_MIX5, d, xd, x) }
    if constexpr (K == MIX7_DPP) { BODYM(I_M7_DPP) }
    if constexpr (K == MIX7_CVT) { BODYM(I_M7_CVT) }
    if constexpr (K == MIX7_MULF64) { BODYM(I_M7_MUL) }
    if constexpr (K == MIX7_SALU) { BODYM(I_M7_SALU) }
    if constexpr (K == MIX7_CND64) { BODYM(I_M7_CND) }
    if constexpr (K == MIX7_MOV) { BODYM(I_M7_MOV) }
    if constexpr (K == MIX7_NOP) { BODYM(I_M7_NOP) }
    if constexpr (K == MIX7_DPP_ROW) { BODYM(I_M7_DPPR) }
    if constexpr (K == MIX7_SWIZZLE) { BODYM(I_M7_SWZ) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (K == MIX7_BPERM) { BODYM(I_M7_BPERM) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (K == MIX7_READLANE) { BODYM(I_M7_RDL) }
    if constexpr (K == MIX7_PERMLANE) { BODYM(I_M7_PERM) }
    if constexpr (K == MIX248_DPP8 || K == MIX1016_DPP8 || K == MIX1016_DPP8_STAGGER) {
#define A64 SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1)
#define OPS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(x) : "v20", "v22"
      if (K == MIX1016_DPP8_STAGGER && it == 0 && ((threadIdx.x >> 8) & 1)) {  // waves 4-7 start half a block later
        asm volatile(A64 A64 A64 A64 OPS);
        asm volatile(A64 A64 A64 A64 OPS);
      }
      asm volatile(A64 A64 A64 SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(D1) OPS);
      if constexpr (K != MIX248_DPP8) {
        asm volatile(A64 A64 A64 A64 OPS);
        asm volatile(A64 A64 A64 A64 OPS);
        asm volatile(A64 A64 A64 A64 OPS);
      }
    }
    if constexpr (K == MIX7_DSREAD) { BODYM2(I_M7_DSR, lds_at) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (K == MIX7_DSWRITE) { BODYM2(I_M7_DSW, lds_at) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (K == MIX7_EXEC) { if (it == 0) asm volatile("v_cmp_eq_u32 vcc, v0, v0" ::: "vcc"); BODYM(I_M7_EXEC) }
    if constexpr (K == MIX7_BUFLOAD) { BODYM2(I_M7_GLD, gptr) asm volatile("s_waitcnt vmcnt(0)"); }
    if constexpr (K == MIX7_WAITCNT) { BODYM(I_M7_WAIT) }
    if constexpr (K == MIX7_CVT32) { BODYM(I_M7_CVT32) }
    if constexpr (K == MIX7_MUL_F32) { BODYM(I_M7_MULF) }
    if constexpr (K == MIX7_FMA_F32) { BODYM(I_M7_FMAF) }
    if constexpr (K == MIX63_BARRIER) {
      asm volatile(SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) X8(A1, 0, 1, 2, 3, 4, 5, 6, 6) "s_barrier\n\t"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(x));
    }
    if constexpr (K == MIX56_DPP8) {
      asm volatile(SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(A1) SEQ8(D1)
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(x) : "v20", "v22");
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
  if (s == 12345.678f) out[threadIdx.x] = s;  // never true in practice; keeps the chains alive
}

template <int K, int NCH>
static double run(int waves_per_simd, float* dout, double clock_hz) {
  const int blocks = 256, threads = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // 100 KB of dynamic LDS: at most one workgroup per compute unit
  hipFuncSetAttribute(reinterpret_cast<const void*>(rate<K, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL((rate<K, NCH>), dim3(blocks), dim3(threads), 100 * 1024, 0, dout, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate<K, NCH>), dim3(blocks), dim3(threads), 100 * 1024, 0, dout, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  const double inst_per_simd = double(waves_per_simd) * ITER * (K == MIX248_DPP8 ? 256.0 : (K == MIX1016_DPP8 || K == MIX1016_DPP8_STAGGER) ? 1024.0 : K >= MIX7_DPP ? 64.0 : K == MIX_CND32_3ADD64 ? 256.0 : K == ADD_LONG_LOOP ? 1024.0 : (K == ADD_SAMEBANK || K == ADD_DIFFBANK || K == ADD_SRC_DST_SAME_BANK) ? 128.0 : K >= MIX_F64_F32 ? 128.0 : 64.0);
  return best * 1e-3 * clock_hz / 4.0 / inst_per_simd;  // quad-cycles per wave-instruction and SIMD
}

template <int K>
static void row(float* dout, double clock_hz) {
  printf("%-34s", kind_name[K]);
  for (int w : {1, 2, 4}) {
    printf("  w%d:", w);
    printf(" %5.2f", run<K, 1>(w, dout, clock_hz));
    printf(" %5.2f", run<K, 2>(w, dout, clock_hz));
    printf(" %5.2f", run<K, 8>(w, dout, clock_hz));
  }
  printf("\n");
  fflush(stdout);
}

template <int K>
static void all_rows(float* dout, double clock_hz) {
  if constexpr (K < NKINDS) {
    row<K>(dout, clock_hz);
    all_rows<K + 1>(dout, clock_hz);
  }
}

// two / four waves per SIMD that belong to DIFFERENT workgroups (256 threads each, one wave per SIMD and workgroup)
template <int K>
static double run_groups(int groups_per_cu, float* dout, double clock_hz) {
  const int lds = 160 * 1024 / groups_per_cu - 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(rate<K, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate<K, 8>), dim3(256 * groups_per_cu), dim3(256), lds, 0, dout, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return best * 1e-3 * clock_hz / 4.0 / (double(groups_per_cu) * ITER * 64.0);
}

int main() {
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double clock_hz = khz * 1e3;
  float* dout;
  if (hipMalloc(&dout, 4096) != hipSuccess) return 1;
  printf("peak clock %.0f MHz; quad-cycles per wave-instruction and SIMD; columns: waves per SIMD x (1, 2, 8 independent chains per wave)\n",
         clock_hz / 1e6);
  printf("v_add_f32, waves of a SIMD from different workgroups (256 threads each): 1 per CU %.2f, 2 per CU %.2f, 4 per CU %.2f\n",
         run_groups<ADD_F32>(1, dout, clock_hz), run_groups<ADD_F32>(2, dout, clock_hz), run_groups<ADD_F32>(4, dout, clock_hz));
  all_rows<0>(dout, clock_hz);
  hipFree(dout);
  return 0;
}
