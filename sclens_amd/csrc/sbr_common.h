// Internal declarations shared by the translation units of the two-stage eigensolver (sbr.hip: band reduction, bulge chase, first
// back-transformation, driver; sbr_q2.hip: second back-transformation).
#pragma once
#include "common.h"

namespace scl {

constexpr int SB = 64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// fp16 pieces of fp32 values, x = hi + lo (22 significant bits): four (K = 16 matrix instruction) or eight (K = 32) per lane
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct SbrHL {
  f16x4 h, l;
};
struct SbrHL8 {
  f16x8 h, l;
};
// x = hi + lo with two packed conversions and one mixed-precision fma per element (lo = x - hi exactly, hi taken as an fp16 operand)
__device__ __forceinline__ SbrHL sbr_split_pk(f32x4 x) {
  const f16x2 h01 = __builtin_convertvector(f32x2{x[0], x[1]}, f16x2);
  const f16x2 h23 = __builtin_convertvector(f32x2{x[2], x[3]}, f16x2);
  const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
  float l0, l1, l2, l3;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(u01), "v"(x[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(u01), "v"(x[1]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(u23), "v"(x[2]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(u23), "v"(x[3]));
  const f16x2 l01 = __builtin_convertvector(f32x2{l0, l1}, f16x2);
  const f16x2 l23 = __builtin_convertvector(f32x2{l2, l3}, f16x2);
  SbrHL o;
  o.h = f16x4{h01[0], h01[1], h23[0], h23[1]};
  o.l = f16x4{l01[0], l01[1], l23[0], l23[1]};
  return o;
}

__device__ __forceinline__ SbrHL8 sbr_cat(const SbrHL& a, const SbrHL& b) {
  SbrHL8 o;
  o.h = __builtin_shufflevector(a.h, b.h, 0, 1, 2, 3, 4, 5, 6, 7);
  o.l = __builtin_shufflevector(a.l, b.l, 0, 1, 2, 3, 4, 5, 6, 7);
  return o;
}

// leading dimension of the reflector store V2[sweep][row] of the bulge chase: 128 spare columns, so that the 64-float run of a reflector
// that starts up to 32 rows past the end (a sweep of a 32-sweep group that has no task k any more) stays inside its own, zero-filled, row
static inline int64_t sbr_ldv2(int64_t n) { return round_up(n, 64) + 128; }
// tasks (64 x 64 block pairs) of sweep s of the chase
__host__ __device__ __forceinline__ int sbr_tasks_of(int64_t s, int64_t n) { return (int)((n - s - 1 + SB - 1) / SB); }

int sbr_ensure_aux(Ctx* ctx);             // the context's second stream and its events, created on first use (sbr.hip)
int sbr_q2_prebuild(Ctx* ctx, int64_t n);  // group data of the second back-transformation on the auxiliary stream (sbr_q2.hip)

}  // namespace scl
