// Shared device/host helpers for the neko_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NEKO_OK 0
#define NEKO_ERR_ARG (-1)
#define NEKO_ERR_UNSUPPORTED (-2)
#define NEKO_ERR_LAUNCH (-3)

typedef uint16_t bf16_t;  // storage type: raw bf16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4;       // ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(16))) float f32x16;     // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define NEKO_WAVE 64

#define NEKO_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return NEKO_ERR_LAUNCH - (int)e__; \
  } while (0)

// ---- bf16 <-> f32 (round to nearest even; NaN preserved) ---------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
// gfx950 has a hardware packed conversion (round-to-nearest-even, NaN-preserving); hipcc lowers a plain
// float->bf16 cast to a ~6-instruction software sequence with a divergent NaN branch, so it is issued explicitly.
// A 2-wide vector convert is what selects v_cvt_pk_bf16_f32 (a scalar cast does not); unlike inline asm the
// compiler then also inserts the VALU->MFMA operand wait states when the result feeds an MFMA directly.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef float f32x2_v __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
  const f32x2_v v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu);
}
// 8 floats -> one MFMA A/B fragment (element j = v[j])
__device__ __forceinline__ bf16x8_v pack8_bf16(const float (&v)[8]) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_v;
  const u32x4_v u = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  return __builtin_bit_cast(bf16x8_v, u);
}

// ---- wave64 reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU and its derivative -- ACT2FN['gelu'] == F.gelu (trajectory_gpt2.py:266)
// erf is evaluated branch-free with Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, i.e. fp32 rounding level):
// libm's erff expands to a divergent multi-branch polynomial that made the GELU epilogues VALU-bound.
//   erf(z) = sign(z) * (1 - (a1 t + ... + a5 t^5) * exp(-z^2)),  t = 1/(1 + p|z|)
// gelu and gelu' share one exponential: exp(-(x/sqrt2)^2) == exp(-x^2/2).
__device__ __forceinline__ void erf_exp_parts(float x, float& erf_z, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn is a 10-instruction IEEE divide
  e = __expf(-z * z);
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  erf_z = copysignf(fmaf(-poly, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_f(float x) {
  float er, e;
  erf_exp_parts(x, er, e);
  return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float er, e;
  erf_exp_parts(x, er, e);
  return fmaf(x * 0.39894228040143267794f, e, 0.5f * (1.0f + er));
}

// Two elements at a time on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of fp32 per issue slot):
// the GEMM epilogues with GELU / GELU' are VALU-bound (256 x 256 results x ~16 slots each per tile against a 24-k-tile
// main loop), and everything in the formulas above except v_rcp / v_exp / the sign transfer packs.  Same series, same
// constants; gelu is rearranged so that the sign never has to be transferred:
//   gelu(x)  = x/2 + |x|/2 * m,              m = 1 - poly(t) e   (erf(|z|))
//   gelu'(x) = 1/2 + copysign(m/2, x) + x e / sqrt(2 pi)
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void erf_parts2(f32x2_v x, f32x2_v ax, float pscale, f32x2_v& poly, f32x2_v& e) {
  const f32x2_v d = __builtin_elementwise_fma(ax, (f32x2_v)(0.70710678118654752440f * 0.3275911f), (f32x2_v)(1.0f));
  f32x2_v t;
  t.x = __builtin_amdgcn_rcpf(d.x);
  t.y = __builtin_amdgcn_rcpf(d.y);
  const f32x2_v a = (x * x) * (-0.5f * 1.44269504088896340736f);          // -(x/sqrt2)^2 * log2(e)
  e.x = __builtin_amdgcn_exp2f(a.x);
  e.y = __builtin_amdgcn_exp2f(a.y);
  poly = __builtin_elementwise_fma(t, (f32x2_v)(1.061405429f * pscale), (f32x2_v)(-1.453152027f * pscale));
  poly = __builtin_elementwise_fma(poly, t, (f32x2_v)(1.421413741f * pscale));
  poly = __builtin_elementwise_fma(poly, t, (f32x2_v)(-0.284496736f * pscale));
  poly = __builtin_elementwise_fma(poly, t, (f32x2_v)(0.254829592f * pscale));
  poly *= t;
}
__device__ __forceinline__ f32x2_v gelu2_f(f32x2_v x) {
  const f32x2_v ax = __builtin_elementwise_abs(x);
  f32x2_v poly, e;
  erf_parts2(x, ax, -1.0f, poly, e);                                      // poly = -(a1 t + ... + a5 t^5)
  const f32x2_v m = __builtin_elementwise_fma(poly, e, (f32x2_v)(1.0f));
  return __builtin_elementwise_fma(ax * 0.5f, m, x * 0.5f);
}
__device__ __forceinline__ f32x2_v gelu_grad2_f(f32x2_v x) {
  const f32x2_v ax = __builtin_elementwise_abs(x);
  f32x2_v poly, e;
  erf_parts2(x, ax, -0.5f, poly, e);
  const f32x2_v hm = __builtin_elementwise_fma(poly, e, (f32x2_v)(0.5f));   // erf(|z|) / 2
  f32x2_v her;
  her.x = copysignf(hm.x, x.x);
  her.y = copysignf(hm.y, x.y);
  return __builtin_elementwise_fma(x * 0.39894228040143267794f, e, her + 0.5f);
}

// gelu(x) and gelu'(x) of a pair from one evaluation of the series (the patch kernels' backward needs both)
__device__ __forceinline__ void gelu_and_grad2_f(f32x2_v x, f32x2_v& g, f32x2_v& gp) {
  const f32x2_v ax = __builtin_elementwise_abs(x);
  f32x2_v poly, e;
  erf_parts2(x, ax, -0.5f, poly, e);
  const f32x2_v hm = __builtin_elementwise_fma(poly, e, (f32x2_v)(0.5f));   // erf(|z|) / 2
  f32x2_v her;
  her.x = copysignf(hm.x, x.x);
  her.y = copysignf(hm.y, x.y);
  const f32x2_v cdf = her + 0.5f;
  g = x * cdf;
  gp = __builtin_elementwise_fma(x * 0.39894228040143267794f, e, cdf);
}

// ---- dropout: counter-based keep decisions (no mask tensor; forward and backward regenerate them) ----------------
// One 32-bit hash word serves FOUR consecutive elements: element idx keeps iff byte (idx & 3) of
// drop_word(idx >> 2, site key) >= thr, thr = round(p*256).  The drop rate is quantised to 1/256 (p = 0.1 -> 26/256)
// and the survivors are scaled by 256/(256-thr), so E[out] = in exactly.  The word is built from full-rate 24-bit
// multiplies (v_mul_u32_u24 / v_mad_u32_u24; a 32-bit v_mul_lo_u32 is quarter rate): 9 VALU ops per 4 elements plus
// a byte extract + compare each, against 13 slots per element for the earlier 2 x mul_lo hash (dropout was +55 % on
// the attention forward).  Attention indexes it in 2-D, see attention.hip.  The site key (per layer / site / step) is
// mixed on the host; tests/test_dropout_gpu.py restates the function in numpy.
__device__ __forceinline__ uint32_t drop_word(uint32_t g, uint32_t key) {
  uint32_t h = g ^ key;
  h = __umul24(h, 0x9E3779u) + __umul24(h >> 8, 0x85EBCBu);   // low 24 bits and bits 8..31: every input bit counts
  h ^= h >> 15;
  h = __umul24(h, 0xC2B2AFu);
  h ^= h >> 16;
  return h;
}
// Device-side key offset for captured (HIP-graph) training steps: kernel arguments are frozen at capture time, so the per-step
// variation of every dropout site comes from ONE uint32 in device memory that the captured step itself advances.  Each kernel
// with a dropout site adds neko_drop_salt() to its site key once, at entry (0 when no salt is registered: eager mode).
// One copy of the pointer per translation unit (the library is built without relocatable device code); neko_set_drop_salt()
// in neko_capi.hip sets them all.
static __device__ const uint32_t* g_neko_drop_salt = nullptr;
__device__ __forceinline__ uint32_t neko_drop_salt() {
  const uint32_t* p = g_neko_drop_salt;
  return p ? *p : 0u;
}
#define NEKO_DEFINE_SALT_SETTER(tag)                                                                              \
  int neko_set_drop_salt_##tag(const uint32_t* p) {                                                               \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_neko_drop_salt), &p, sizeof(p)) == hipSuccess ? NEKO_OK : NEKO_ERR_LAUNCH; \
  }
__device__ __forceinline__ bool drop_byte_keep(uint32_t word, int i, uint32_t thr) { return ((word >> (8 * i)) & 0xffu) >= thr; }
__device__ __forceinline__ bool drop_keep(uint32_t idx, uint32_t key, uint32_t thr) {
  return ((drop_word(idx >> 2, key) >> (8 * (idx & 3u))) & 0xffu) >= thr;
}
// v[0..3] = elements base .. base+3: masked and scaled in place (one word when base is a multiple of 4)
__device__ __forceinline__ void drop4(float (&v)[4], uint32_t base, uint32_t key, uint32_t thr, float scale) {
  if ((base & 3u) == 0) {
    const uint32_t w = drop_word(base >> 2, key);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = drop_byte_keep(w, e, thr) ? v[e] * scale : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = drop_keep(base + e, key, thr) ? v[e] * scale : 0.f;
  }
}

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b%8): give every XCD a
// contiguous range of logical tile ids so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
