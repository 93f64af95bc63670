// Shared device helpers for the protstruc geometry kernels (gfx950 only).
//
// Every translation unit is compiled with -ffp-contract=off: the reference's
// arithmetic is a chain of separate ATen / numpy multiplies, adds and
// subtracts, and fused multiply-adds change results that must be *exact*
// zeros there (e.g. the diagonal of pairwise_dihedrals, SURVEY hard part 4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tuple>
#include <utility>

#define PS_WAVE 64

struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 div3(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }

// (x*y).sum(-1): products first, then adds (geometry.py:24-26).  ATen's sum
// accumulates from +0, so a dot product whose three terms are all -0 is +0 in
// the reference; that sign decides atan2(0, x) on degenerate (i == j) pairs, so
// the leading `0 +` is part of the contract (it is exact for every other input).
__device__ __forceinline__ float dot3(f3 a, f3 b) {
    float px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
    return ((0.0f + px) + py) + pz;
}

// Correctly rounded sqrt for x >= 0 without the subnormal pre-scaling of the
// library routine: v_sqrt_f32 is within 1 ulp, so the answer is s-1ulp, s or
// s+1ulp and two exact fma residuals pick it.  0, inf and NaN fall through
// unchanged (every comparison with a NaN residual is false).  Checked against
// sqrtf over every float in [0, +inf] (tools/microbench/sqrt_check.hip): equal
// for every x >= 4.6e-32; below that (atoms closer than 2e-16) the residuals
// underflow and the result can be 1 ulp off.  Kept as the cross-check of
// sqrt_rn_mk below, which is what the kernels use.
__device__ __forceinline__ float sqrt_rn_pos(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __uint_as_float(__float_as_uint(s) - 1u);
    const float hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_lo = __builtin_fmaf(-lo, s, x);
    const float r_hi = __builtin_fmaf(-hi, s, x);
    float r = s;
    r = (r_lo <= 0.0f) ? lo : r;
    r = (r_hi > 0.0f) ? hi : r;
    return r;
}

// The same correctly rounded result from v_rsq_f32: one coupled Newton step for g ~ sqrt(x) and h ~ 1/(2 sqrt(x)),
// then the exact-residual correction g + (x - g*g) * h (fma).  Every operation is a mul or an fma, so two elements
// share one v_pk_* instruction: 4 + 3.5 + 2 issue slots per element against 4 + 9 for sqrt_rn_pos.  Zero, subnormal
// and +inf inputs are returned as x (sqrt(0) = 0 and sqrt(inf) = inf exactly; a subnormal squared distance means
// atoms closer than 1e-19, error < 1.1e-19).  Checked against sqrtf over every float in [0, +inf]
// (tools/microbench/sqrt_check.hip): equal for every x >= 2.0e-31 (atoms further apart than 4.5e-16), at most
// 1 ulp off below.  In K1's pattern kernel this takes the inner loop from 66 to 50 VALU instructions per 16-byte
// slot, which is what lets it reach its store-only time (profiles/r01_k1_ab_sqrt_mk.log).
__device__ __forceinline__ float sqrt_rn_mk(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    return __builtin_amdgcn_classf(x, 0x2F0) ? x : g;   // +-0, +-subnormal, +inf
}

// |a - b| exactly as K1 evaluates it (protstruc.py:477-479): differences, squares, two adds, sqrt
__device__ __forceinline__ float dist3(f3 a, f3 b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    return sqrt_rn_mk((sx + sy) + sz);
}

// the same with K1's choice of square root (ps_k1_config.exact_sqrt): hardware v_sqrt_f32 or correctly rounded
template <bool EXACT>
__device__ __forceinline__ float dist3_t(f3 a, f3 b) {
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    const float x = (sx + sy) + sz;
    return EXACT ? sqrt_rn_mk(x) : __builtin_amdgcn_sqrtf(x);
}

// x.norm(dim=-1) (geometry.py:29-31); correctly rounded sqrt in half the instructions of the library routine
// (identical result unless the squared norm is subnormal, i.e. |a| < 1e-19)
__device__ __forceinline__ float norm3(f3 a) { return sqrt_rn_mk(dot3(a, a)); }

// np.cross component order: u1*v2 - u2*v1, ... two products then one subtract
__device__ __forceinline__ f3 cross3(f3 u, f3 v) {
    f3 r;
    r.x = u.y * v.z - u.z * v.y;
    r.y = u.z * v.x - u.x * v.z;
    r.z = u.x * v.y - u.y * v.x;
    return r;
}

// atan2 for the torsion kernels: odd minimax polynomial of degree 17 on [0,1] (max error 1.0e-7 = 1.7 ulp in
// fp32, fitted and checked in fp32 emulation over 2e6 points), argument reduction by min/max and the usual
// quadrant fix-ups.  IEEE special cases are kept: signed zeros (atan2(+0,-0) = pi, atan2(-0,+0) = -0), NaN
// propagation, inf/inf.  About half the instructions of the library routine; K3 is VALU-bound, so this is
// where its time goes.  The reference's own np.arctan2 / libm results differ from each other by similar amounts.
__device__ __forceinline__ float atan2_ps(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mn * __builtin_amdgcn_rcpf(mx);
    a = (mx == 0.0f) ? 0.0f : a;                          // atan2(+-0, +-0)
    a = (mn == __builtin_huge_valf()) ? 1.0f : a;         // atan2(+-inf, +-inf)
    const float s = a * a;
    float p = 0.0028340641874819994f;
    p = __builtin_fmaf(p, s, -0.016005029901862144f);
    p = __builtin_fmaf(p, s, 0.042587608098983765f);
    p = __builtin_fmaf(p, s, -0.07495445758104324f);
    p = __builtin_fmaf(p, s, 0.10636754333972931f);
    p = __builtin_fmaf(p, s, -0.14202570915222168f);
    p = __builtin_fmaf(p, s, 0.19992484152317047f);
    p = __builtin_fmaf(p, s, -0.3333306610584259f);
    p = __builtin_fmaf(p, s, 1.0f);
    float r = a * p;
    r = (ay > ax) ? (1.5707963267948966f - r) : r;
    r = (__float_as_uint(x) >> 31) ? (3.141592653589793f - r) : r;
    r = (x != x || y != y) ? __builtin_nanf("") : r;
    return copysignf(r, y);
}

// acos for the planar-angle kernels: acos(|x|) = sqrt(1 - |x|) * P(|x|) with the degree-7 polynomial of Abramowitz &
// Stegun 4.4.46 (|error| <= 2e-8 in exact arithmetic; 4.3e-7 absolute = 1 ulp of pi in fp32, checked over 4.2e6 points
// incl. 1e-12 from +-1 -- the same size as the library routine's), reflected for negative arguments.  |x| > 1 gives NaN
// through the square root of a negative number, as acos without a clamp does in the reference (geometry.py:64-71); NaN
// propagates.  14 instructions instead of the library's ~40: K3's planar-angle path is VALU-issue bound.
__device__ __forceinline__ float acos_ps(float x) {
    const float ax = fabsf(x);
    float p = -0.0012624911f;
    p = __builtin_fmaf(p, ax, 0.0066700901f);
    p = __builtin_fmaf(p, ax, -0.0170881256f);
    p = __builtin_fmaf(p, ax, 0.0308918810f);
    p = __builtin_fmaf(p, ax, -0.0501743046f);
    p = __builtin_fmaf(p, ax, 0.0889789874f);
    p = __builtin_fmaf(p, ax, -0.2145988016f);
    p = __builtin_fmaf(p, ax, 1.5707963050f);
    const float r = p * __builtin_amdgcn_sqrtf(1.0f - ax);
    return (__float_as_uint(x) >> 31) ? (3.141592653589793f - r) : r;
}

// geometry.dihedral (geometry.py:108-124), with the 1e-5 parity tolerance spent on ONE algebraic step (round 3):
//   reference:  n1 = b0 x b1,  n2 = b2 x b1,  x = n1 . n2,  y = ((n1 x n2) . b1) / |b1|
//   here:       n1, n2, x bit for bit as the reference;  (n1 x n2) = -(n1 . b2) b1 exactly in real arithmetic, so
//               y = -(n1 . b2) |b1| and, atan2 being invariant under a common positive factor,
//               atan2(y, x) = atan2(n1 . (c - d), x / |b1|):
// one cross product (9 of ~47 flops) fewer and no square root, only v_rsq_f32.  What must stay EXACT stays exact:
//   * c - d is the exact negative of the reference's b2 = d - c, and n2 = b1 x (c - d) has the reference's two products
//     per component and their difference, i.e. the same bits as b2 x b1;
//   * on the diagonal of pairwise_dihedrals (i == j) n1 or n2 is exactly 0, both dot products start from +0 (dot3), so
//     y = +0 and x = +0 * rsq = +0 and the result is +0.0 with no sign bit, as in the reference (golden G3);
//   * b1 = 0 gives x = 0 * inf = NaN like the reference's 0 / 0; NaN coordinates propagate.
// Everywhere else the result differs from the reference's by rounding only (max 2.4e-7 off-diagonal at unit scale
// before and after, tools/k3_error_stats.py; the conditioning gates of tests/test_gpu_parity.py hold it to the oracle
// and to fp64).
__device__ __forceinline__ float dihedral4(f3 a, f3 b, f3 c, f3 d) {
    f3 b0 = sub3(a, b);
    f3 b1 = sub3(c, b);
    f3 b2n = sub3(c, d);
    f3 n1 = cross3(b0, b1);
    f3 n2 = cross3(b1, b2n);
    float x = dot3(n1, n2) * __builtin_amdgcn_rsqf(dot3(b1, b1));
    float y = dot3(n1, b2n);
    return atan2_ps(y, x);
}

// ---- the FAITHFUL forms (round 4): geometry.dihedral / geometry.angle op for op in the reference's order ----
// What `exact_angles` of ps_pairwise_angles_f32 / ps_inter_residue_geometry_f32 selects, as `exact_sqrt` does for K1: three
// cross products, y divided by |b1| (correctly rounded square root, IEEE division), atan2 / acos in the device library's
// arithmetic (<= 2 ulp; the reference's np.arctan2 / torch.arccos are libm-grade as well) -- no algebraic rewriting, no
// reciprocal square roots, no polynomials of this file.
//
// atan2_lib / acos_lib (round 5) ARE the device library's atan2f / acosf (ROCm 7.2 ocml: __ocml_atan2_f32 with its
// __ocmlpriv_atanred_f32, __ocml_acos_f32; denormals on, finite-only off), written out instruction for instruction as hipcc
// -O3 emits a call of them on gfx950 -- read off ocml.bc's IR and the ISA of a one-line kernel -- instead of being called.
// Why not call them: the library asks for its v / u division with `!fpmath 2.5 ulp`, and whether the compiler then emits
// the cheap form (frexp mantissas, v_rcp_f32, v_ldexp_f32) or a full IEEE division turned out to depend on the CALL SITE
// (inside a loop body: the cheap form; hoisted out of a loop whose operands are all loop-invariant -- K3 with every point
// from the column residue -- the IEEE form): two kernels calling atan2f on the same arguments disagreed in the last bit on
// 13 % of them.  Written with builtins the sequence is the same everywhere; it is the cheap form, which is what a plain
// call gives, and tools/microbench/libm_identity.hip holds it to such calls over 2^32 argument pairs (0 mismatches,
// profiles/r05_libm_identity.log).
__device__ __forceinline__ float atan2_lib(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float v = __builtin_fminf(ax, ay), u = __builtin_fmaxf(ax, ay);
    const float m = __builtin_amdgcn_frexp_mantf(v) * __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(u));
    const float q = __builtin_amdgcn_ldexpf(m, __builtin_amdgcn_frexp_expf(v) - __builtin_amdgcn_frexp_expf(u));
    const float t = q * q;
    float p = __builtin_fmaf(t, __uint_as_float(0x3b2d2a58u), __uint_as_float(0xbc7a590cu));
    p = __builtin_fmaf(t, p, __uint_as_float(0x3d29fb3fu));
    p = __builtin_fmaf(t, p, __uint_as_float(0xbd97d4d7u));
    p = __builtin_fmaf(t, p, __uint_as_float(0x3dd931b2u));
    p = __builtin_fmaf(t, p, __uint_as_float(0xbe1160e6u));
    p = __builtin_fmaf(t, p, __uint_as_float(0x3e4cb8bfu));
    p = __builtin_fmaf(t, p, __uint_as_float(0xbeaaaa62u));
    p = t * p;
    float a = __builtin_fmaf(q, p, q);
    const float t1 = __uint_as_float(0x3fc90fdbu) - a;
    a = ay > ax ? t1 : a;
    const float t2 = __uint_as_float(0x40490fdbu) - a;
    a = x < 0.0f ? t2 : a;
    const float t3 = ((int)__float_as_uint(x) < 0) ? __uint_as_float(0x40490fdbu) : 0.0f;
    a = (y == 0.0f) ? t3 : a;
    const float t4 = (x < 0.0f) ? __uint_as_float(0x4016cbe4u) : __uint_as_float(0x3f490fdbu);
    a = (__builtin_isinf(x) && __builtin_isinf(y)) ? t4 : a;
    a = (x != x || y != y) ? __uint_as_float(0x7fc00000u) : a;
    return copysignf(a, y);
}

__device__ __forceinline__ float acos_lib(float x) {
    const float ax = fabsf(x);
    const float h = __builtin_fmaf(ax, -0.5f, 0.5f), s = x * x;
    const bool big = ax > 0.5f;
    const float w = big ? h : s;
    float p = __builtin_fmaf(w, __uint_as_float(0x3d1c21a7u), __uint_as_float(0x3c5fc5dau));
    p = __builtin_fmaf(w, p, __uint_as_float(0x3d034c3cu));
    p = __builtin_fmaf(w, p, __uint_as_float(0x3d3641b1u));
    p = __builtin_fmaf(w, p, __uint_as_float(0x3d999bc8u));
    p = __builtin_fmaf(w, p, __uint_as_float(0x3e2aaaacu));
    const float z = w * p;
    const float sq = __builtin_amdgcn_sqrtf(w);
    const float g = __builtin_fmaf(sq, z, sq);
    const float g2 = g + g;
    const float neg = __uint_as_float(0x40490fdbu) - g2;
    const float sm = __uint_as_float(0x3fc90fdbu) - __builtin_fmaf(x, z, x);
    return big ? (x < 0.0f ? neg : g2) : sm;
}

__device__ __forceinline__ float dihedral4_ref(f3 a, f3 b, f3 c, f3 d) {
    const f3 b0 = sub3(a, b), b1 = sub3(c, b), b2 = sub3(d, c);
    const f3 n1 = cross3(b0, b1);          // np.cross(b0, b1)
    const f3 n2 = cross3(b2, b1);          // np.cross(b2, b1)
    const f3 m = cross3(n1, n2);
    const float x = dot3(n1, n2);
    const float y = dot3(m, b1) / norm3(b1);   // an IEEE division: correctly rounded, hence the same bits however it is lowered
    return atan2_lib(y, x);
}

__device__ __forceinline__ float angle3_ref(f3 a, f3 b, f3 c) {
    const f3 ba = sub3(a, b), bc = sub3(c, b);
    const float cosine = dot3(ba, bc) / (norm3(ba) * norm3(bc));
    return acos_lib(cosine);               // no clamp, as geometry.py:64-71
}

// Dot product of the FAST pairwise forms (round 5): fma(a.z, b.z, fma(a.y, b.y, fma(a.x, b.x, +0))) -- three instructions
// instead of five (K3's time is the vector unit's).  The first fma keeps dot3's `0 +`: a sum of -0 terms is still +0, which is
// what makes the (i == j) diagonal of pairwise_dihedrals exactly +0.0; a sum of exact zeros stays an exact zero.  Elsewhere
// it differs from the reference's products-then-adds by rounding only (one rounding instead of three: closer to the
// real dot product).  The cross products keep their unfused form: a * b - a * b has to cancel exactly on the diagonal.  The
// faithful forms (dihedral4_ref, angle3_ref) do not use it.
__device__ __forceinline__ float dot3_fast(f3 a, f3 b) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, __builtin_fmaf(a.x, b.x, 0.0f)));
}

// atan2 of the PAIRWISE kernels (K3 and the fused featuriser; K2 and the pointwise entry keep atan2_ps with every IEEE
// special case).  Same polynomial, same quadrant logic, same signed-zero behaviour (atan2(+0, +0) = +0, atan2(+0, -0) =
// pi); what is dropped are six per-element compare / select instructions that only matter for infinite arguments:
//   * max(|x|, |y|) is clamped below at FLT_MIN inside the same v_max3_f32, which makes 0 / 0 come out as 0 without the
//     separate (mx == 0) select;
//   * NaN: v_min / v_max drop NaNs, so a NaN x is re-injected with one fused multiply-add by zero (instead of two
//     compares, an or and a select).  y needs none: in dihedral4_k3 y = n1 . (c - d) is NaN only if a coordinate is, and
//     then x = (n1 . n2) * rsq is NaN as well; x alone is NaN when b1 = 0 (0 * inf), the reference's 0 / 0;
//   * atan2(+-inf, +-inf) and atan2(finite, +-inf) come out NaN instead of multiples of pi / 4.  x and y are dot
//     products of cross products of coordinate differences: they overflow only for coordinates beyond 4e9 A.
__device__ __forceinline__ float atan2_k3(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1.17549435e-38f), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);
    const float s = a * a;
    float p = 0.0028340641874819994f;
    p = __builtin_fmaf(p, s, -0.016005029901862144f);
    p = __builtin_fmaf(p, s, 0.042587608098983765f);
    p = __builtin_fmaf(p, s, -0.07495445758104324f);
    p = __builtin_fmaf(p, s, 0.10636754333972931f);
    p = __builtin_fmaf(p, s, -0.14202570915222168f);
    p = __builtin_fmaf(p, s, 0.19992484152317047f);
    p = __builtin_fmaf(p, s, -0.3333306610584259f);
    p = __builtin_fmaf(p, s, 1.0f);
    float r = a * p;
    r = (ay > ax) ? (1.5707963267948966f - r) : r;
    r = (__float_as_uint(x) >> 31) ? (3.141592653589793f - r) : r;
    r = __builtin_fmaf(x, 0.0f, r);   // r >= +0 here: adding +-0 leaves it, a NaN (or infinite) x poisons it
    return copysignf(r, y);
}

// dihedral4 with atan2_k3 and the three-instruction dot products (dot3_fast)
__device__ __forceinline__ float dihedral4_k3(f3 a, f3 b, f3 c, f3 d) {
    f3 b0 = sub3(a, b);
    f3 b1 = sub3(c, b);
    f3 b2n = sub3(c, d);
    f3 n1 = cross3(b0, b1);
    f3 n2 = cross3(b1, b2n);
    const float nn = dot3_fast(b1, b1);
    float x = dot3_fast(n1, n2) * __builtin_amdgcn_rsqf(nn);
    float y = dot3_fast(n1, b2n);
    return atan2_k3(y, x);
}

// ---- two problems per lane: the same arithmetic on float2, which gfx950 issues as packed v_pk_* instructions ----
// (K3 is VALU-issue bound; element for element the operations and their order are those of the scalar code above,
// so the results are bit-identical.)
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct f3v {
    f32x2 x, y, z;
};

__device__ __forceinline__ f3v mk3v(f3 a, f3 b) { return f3v{f32x2{a.x, b.x}, f32x2{a.y, b.y}, f32x2{a.z, b.z}}; }
__device__ __forceinline__ f3v sub3v(f3v a, f3v b) { return f3v{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f32x2 dot3v(f3v a, f3v b) {
    const f32x2 px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
    return ((f32x2{0.0f, 0.0f} + px) + py) + pz;
}
__device__ __forceinline__ f3v cross3v(f3v u, f3v v) {
    f3v r;
    r.x = u.y * v.z - u.z * v.y;
    r.y = u.z * v.x - u.x * v.z;
    r.z = u.x * v.y - u.y * v.x;
    return r;
}
// dot3_fast on both halves
__device__ __forceinline__ f32x2 dot3v_fast(f3v a, f3v b) {
    return __builtin_elementwise_fma(a.z, b.z, __builtin_elementwise_fma(a.y, b.y, __builtin_elementwise_fma(a.x, b.x, f32x2{0.0f, 0.0f})));
}
// atan2_ps on both halves: the polynomial, the squares and the quadrant subtractions as packed mul / fma / add
// (v_pk_*_f32 issue at the scalar rate, so every packed instruction retires two elements' worth); abs / min / max /
// rcp / selects have no packed fp32 form and stay per element.  Same operations in the same order per element as
// atan2_ps, hence the same bits (K3's odd last row and the fused featuriser use the scalar routine).
__device__ __forceinline__ f32x2 atan2_ps_v(f32x2 y, f32x2 x) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)}, ay = {fabsf(y.x), fabsf(y.y)};
    const f32x2 mx = {fmaxf(ax.x, ay.x), fmaxf(ax.y, ay.y)}, mn = {fminf(ax.x, ay.x), fminf(ax.y, ay.y)};
    f32x2 a = mn * f32x2{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
    a.x = (mx.x == 0.0f) ? 0.0f : a.x;
    a.y = (mx.y == 0.0f) ? 0.0f : a.y;
    a.x = (mn.x == __builtin_huge_valf()) ? 1.0f : a.x;
    a.y = (mn.y == __builtin_huge_valf()) ? 1.0f : a.y;
    const f32x2 s = a * a;
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 p = k2(0.0028340641874819994f);
    p = __builtin_elementwise_fma(p, s, k2(-0.016005029901862144f));
    p = __builtin_elementwise_fma(p, s, k2(0.042587608098983765f));
    p = __builtin_elementwise_fma(p, s, k2(-0.07495445758104324f));
    p = __builtin_elementwise_fma(p, s, k2(0.10636754333972931f));
    p = __builtin_elementwise_fma(p, s, k2(-0.14202570915222168f));
    p = __builtin_elementwise_fma(p, s, k2(0.19992484152317047f));
    p = __builtin_elementwise_fma(p, s, k2(-0.3333306610584259f));
    p = __builtin_elementwise_fma(p, s, k2(1.0f));
    f32x2 r = a * p;
    const f32x2 rq = k2(1.5707963267948966f) - r;
    r.x = (ay.x > ax.x) ? rq.x : r.x;
    r.y = (ay.y > ax.y) ? rq.y : r.y;
    const f32x2 rh = k2(3.141592653589793f) - r;
    r.x = (__float_as_uint(x.x) >> 31) ? rh.x : r.x;
    r.y = (__float_as_uint(x.y) >> 31) ? rh.y : r.y;
    r.x = (x.x != x.x || y.x != y.x) ? __builtin_nanf("") : r.x;
    r.y = (x.y != x.y || y.y != y.y) ? __builtin_nanf("") : r.y;
    return f32x2{copysignf(r.x, y.x), copysignf(r.y, y.y)};
}

__device__ __forceinline__ f32x2 dihedral4v(f3v a, f3v b, f3v c, f3v d) {
    const f3v b0 = sub3v(a, b), b1 = sub3v(c, b), b2n = sub3v(c, d);
    const f3v n1 = cross3v(b0, b1), n2 = cross3v(b1, b2n);
    const f32x2 nn = dot3v(b1, b1);
    const f32x2 x = dot3v(n1, n2) * f32x2{__builtin_amdgcn_rsqf(nn.x), __builtin_amdgcn_rsqf(nn.y)};
    const f32x2 y = dot3v(n1, b2n);
    return atan2_ps_v(y, x);
}

// atan2_k3 / dihedral4_k3 on both halves (same operations in the same order per element, hence the same bits)
__device__ __forceinline__ f32x2 atan2_k3_v(f32x2 y, f32x2 x) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)}, ay = {fabsf(y.x), fabsf(y.y)};
    const f32x2 mx = {fmaxf(fmaxf(ax.x, ay.x), 1.17549435e-38f), fmaxf(fmaxf(ax.y, ay.y), 1.17549435e-38f)};
    const f32x2 mn = {fminf(ax.x, ay.x), fminf(ax.y, ay.y)};
    const f32x2 a = mn * f32x2{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
    const f32x2 s = a * a;
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 p = k2(0.0028340641874819994f);
    p = __builtin_elementwise_fma(p, s, k2(-0.016005029901862144f));
    p = __builtin_elementwise_fma(p, s, k2(0.042587608098983765f));
    p = __builtin_elementwise_fma(p, s, k2(-0.07495445758104324f));
    p = __builtin_elementwise_fma(p, s, k2(0.10636754333972931f));
    p = __builtin_elementwise_fma(p, s, k2(-0.14202570915222168f));
    p = __builtin_elementwise_fma(p, s, k2(0.19992484152317047f));
    p = __builtin_elementwise_fma(p, s, k2(-0.3333306610584259f));
    p = __builtin_elementwise_fma(p, s, k2(1.0f));
    f32x2 r = a * p;
    const f32x2 rq = k2(1.5707963267948966f) - r;
    r.x = (ay.x > ax.x) ? rq.x : r.x;
    r.y = (ay.y > ax.y) ? rq.y : r.y;
    const f32x2 rh = k2(3.141592653589793f) - r;
    r.x = (__float_as_uint(x.x) >> 31) ? rh.x : r.x;
    r.y = (__float_as_uint(x.y) >> 31) ? rh.y : r.y;
    r = __builtin_elementwise_fma(x, k2(0.0f), r);
    return f32x2{copysignf(r.x, y.x), copysignf(r.y, y.y)};
}

__device__ __forceinline__ f32x2 dihedral4v_k3(f3v a, f3v b, f3v c, f3v d) {
    const f3v b0 = sub3v(a, b), b1 = sub3v(c, b), b2n = sub3v(c, d);
    const f3v n1 = cross3v(b0, b1), n2 = cross3v(b1, b2n);
    const f32x2 nn = dot3v_fast(b1, b1);
    const f32x2 x = dot3v_fast(n1, n2) * f32x2{__builtin_amdgcn_rsqf(nn.x), __builtin_amdgcn_rsqf(nn.y)};
    const f32x2 y = dot3v_fast(n1, b2n);
    return atan2_k3_v(y, x);
}

// ---- NC column residues per lane: the same arithmetic STEP BY STEP across the columns ----
// (round 4)  K3's time is the VALU's (tools/microbench/k3_valu_floor.hip), and a dependent v_pk_fma_f32 costs ~11 cycles
// where an independent one costs 4: left to itself the compiler evaluates one column's Horner chain after the other with
// an s_nop between the links.  Written across the columns, with a scheduling barrier after every link, the NC chains
// interleave and the loop needs no s_nop at all: 0.60 -> 0.53 us per (4 columns x 2 rows) trip on one SIMD.  Per element
// the operations and their order are those of atan2_k3 / acos_ps, hence the same bits.
template <int NC>
__device__ __forceinline__ void atan2_k3_vn(const f32x2 (&y)[NC], const f32x2 (&x)[NC], f32x2 (&r)[NC]) {
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 ax[NC], ay[NC], a[NC], s[NC], p[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        ax[c] = f32x2{fabsf(x[c].x), fabsf(x[c].y)};
        ay[c] = f32x2{fabsf(y[c].x), fabsf(y[c].y)};
        const f32x2 mx = {fmaxf(fmaxf(ax[c].x, ay[c].x), 1.17549435e-38f), fmaxf(fmaxf(ax[c].y, ay[c].y), 1.17549435e-38f)};
        const f32x2 mn = {fminf(ax[c].x, ay[c].x), fminf(ax[c].y, ay[c].y)};
        a[c] = mn * f32x2{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] = a[c] * a[c];
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(k2(0.0028340641874819994f), s[c], k2(-0.016005029901862144f));
    constexpr float co[7] = {0.042587608098983765f, -0.07495445758104324f, 0.10636754333972931f, -0.14202570915222168f,
                             0.19992484152317047f, -0.3333306610584259f, 1.0f};
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(p[c], s[c], k2(co[t]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) r[c] = a[c] * p[c];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x2 rq = k2(1.5707963267948966f) - r[c];
        r[c].x = (ay[c].x > ax[c].x) ? rq.x : r[c].x;
        r[c].y = (ay[c].y > ax[c].y) ? rq.y : r[c].y;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x2 rh = k2(3.141592653589793f) - r[c];
        r[c].x = (__float_as_uint(x[c].x) >> 31) ? rh.x : r[c].x;
        r[c].y = (__float_as_uint(x[c].y) >> 31) ? rh.y : r[c].y;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        r[c] = __builtin_elementwise_fma(x[c], k2(0.0f), r[c]);
        r[c] = f32x2{copysignf(r[c].x, y[c].x), copysignf(r[c].y, y[c].y)};
    }
}

// dot3v with the leading `0 +` folded into the first product: fma(a, b, +0) rounds a * b once and adds +0, which is
// bit for bit (0 + a * b) -- including the -0 -> +0 case the `0 +` exists for -- in one instruction instead of two.
// (The faithful forms' dot product: products first, then adds, as the reference.)
__device__ __forceinline__ f32x2 dot3v_f(f3v a, f3v b) {
    const f32x2 px = __builtin_elementwise_fma(a.x, b.x, f32x2{0.0f, 0.0f}), py = a.y * b.y, pz = a.z * b.z;
    return (px + py) + pz;
}

// dihedral4v_k3 for NC columns: the cross / dot part per column (independent work), then the NC atan2 chains together
template <int NC>
__device__ __forceinline__ void dihedral4v_k3_n(const f3v (&a)[NC], const f3v (&b)[NC], const f3v (&c)[NC],
                                                const f3v (&d)[NC], f32x2 (&out)[NC]) {
    f32x2 x[NC], y[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const f3v b0 = sub3v(a[q], b[q]), b1 = sub3v(c[q], b[q]), b2n = sub3v(c[q], d[q]);
        const f3v n1 = cross3v(b0, b1), n2 = cross3v(b1, b2n);
        const f32x2 nn = dot3v_fast(b1, b1);
        x[q] = dot3v_fast(n1, n2) * f32x2{__builtin_amdgcn_rsqf(nn.x), __builtin_amdgcn_rsqf(nn.y)};
        y[q] = dot3v_fast(n1, b2n);
    }
    atan2_k3_vn<NC>(y, x, out);
}

// acos_ps / angle3v for NC columns (same operations in the same order per element)
template <int NC>
__device__ __forceinline__ void acos_ps_vn(const f32x2 (&x)[NC], f32x2 (&r)[NC]) {
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 ax[NC], p[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) ax[c] = f32x2{fabsf(x[c].x), fabsf(x[c].y)};
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(k2(-0.0012624911f), ax[c], k2(0.0066700901f));
    constexpr float co[6] = {-0.0170881256f, 0.0308918810f, -0.0501743046f, 0.0889789874f, -0.2145988016f, 1.5707963050f};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(p[c], ax[c], k2(co[t]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x2 t = k2(1.0f) - ax[c];
        const f32x2 v = p[c] * f32x2{__builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y)};
        const f32x2 rh = k2(3.141592653589793f) - v;
        r[c] = f32x2{(__float_as_uint(x[c].x) >> 31) ? rh.x : v.x, (__float_as_uint(x[c].y) >> 31) ? rh.y : v.y};
    }
}

// acos_ps on both halves (same operations in the same order per element)
__device__ __forceinline__ f32x2 acos_ps_v(f32x2 x) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 p = k2(-0.0012624911f);
    p = __builtin_elementwise_fma(p, ax, k2(0.0066700901f));
    p = __builtin_elementwise_fma(p, ax, k2(-0.0170881256f));
    p = __builtin_elementwise_fma(p, ax, k2(0.0308918810f));
    p = __builtin_elementwise_fma(p, ax, k2(-0.0501743046f));
    p = __builtin_elementwise_fma(p, ax, k2(0.0889789874f));
    p = __builtin_elementwise_fma(p, ax, k2(-0.2145988016f));
    p = __builtin_elementwise_fma(p, ax, k2(1.5707963050f));
    const f32x2 t = k2(1.0f) - ax;
    const f32x2 r = p * f32x2{__builtin_amdgcn_sqrtf(t.x), __builtin_amdgcn_sqrtf(t.y)};
    const f32x2 rh = k2(3.141592653589793f) - r;
    return f32x2{(__float_as_uint(x.x) >> 31) ? rh.x : r.x, (__float_as_uint(x.y) >> 31) ? rh.y : r.y};
}

// sqrt_rn_mk on both halves: v_rsq_f32 and the class test per element, the Newton / residual steps as packed fma
// (element for element the operations of the scalar routine, hence the same bits)
__device__ __forceinline__ f32x2 sqrt_rn_mk_v(f32x2 x) {
    const f32x2 y = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    f32x2 g = x * y;
    f32x2 h = f32x2{0.5f, 0.5f} * y;
    const f32x2 r = __builtin_elementwise_fma(-h, g, f32x2{0.5f, 0.5f});
    g = __builtin_elementwise_fma(g, r, g);
    h = __builtin_elementwise_fma(h, r, h);
    const f32x2 d = __builtin_elementwise_fma(-g, g, x);
    g = __builtin_elementwise_fma(d, h, g);
    return f32x2{__builtin_amdgcn_classf(x.x, 0x2F0) ? x.x : g.x, __builtin_amdgcn_classf(x.y, 0x2F0) ? x.y : g.y};
}

// dist3 / angle3 for two problems per lane (same operations in the same order as the scalar functions below / above)
__device__ __forceinline__ f32x2 dist3v(f3v a, f3v b) {
    const f32x2 dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    const f32x2 sx = dx * dx, sy = dy * dy, sz = dz * dz;
    return sqrt_rn_mk_v((sx + sy) + sz);
}

// geometry.angle (geometry.py:64-71): no clamp before acos.  The reference divides the dot product by the product of
// two norms; here cos = ((ba . bc) * rsq(ba . ba)) * rsq(bc . bc) -- two v_rsq_f32 (1 ulp each) instead of two correctly
// rounded square roots and an IEEE divide (4 instructions instead of ~29) -- and acos_ps above.  Each arm is scaled away
// before the next factor comes in (round 5; rounds 3-4 multiplied the two squared lengths first, which overflowed for
// arms beyond 4e9 and underflowed below 1e-10): any arm between 1e-19 and 1e19 is safe, as in the reference.  A
// zero-length arm (the diagonal of pairwise_planar_angles) is 0 * rsq(0) = 0 * inf = NaN like the reference's 0 / 0.
template <bool EXACT>
__device__ __forceinline__ f32x2 dist3v_t(f3v a, f3v b) {
    const f32x2 dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    const f32x2 sx = dx * dx, sy = dy * dy, sz = dz * dz;
    const f32x2 x = (sx + sy) + sz;
    if (EXACT) return sqrt_rn_mk_v(x);
    return f32x2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
}

__device__ __forceinline__ f32x2 angle3v(f3v a, f3v b, f3v c) {
    const f3v ba = sub3v(a, b), bc = sub3v(c, b);
    const f32x2 num = dot3v_fast(ba, bc);
    const f32x2 na = dot3v_fast(ba, ba), nb = dot3v_fast(bc, bc);
    return acos_ps_v((num * f32x2{__builtin_amdgcn_rsqf(na.x), __builtin_amdgcn_rsqf(na.y)}) * f32x2{__builtin_amdgcn_rsqf(nb.x), __builtin_amdgcn_rsqf(nb.y)});
}

template <int NC>
__device__ __forceinline__ void angle3v_n(const f3v (&a)[NC], const f3v (&b)[NC], const f3v (&c)[NC], f32x2 (&out)[NC]) {
    f32x2 cs[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const f3v ba = sub3v(a[q], b[q]), bc = sub3v(c[q], b[q]);
        const f32x2 num = dot3v_fast(ba, bc);
        const f32x2 na = dot3v_fast(ba, ba), nb = dot3v_fast(bc, bc);
        cs[q] = (num * f32x2{__builtin_amdgcn_rsqf(na.x), __builtin_amdgcn_rsqf(na.y)}) * f32x2{__builtin_amdgcn_rsqf(nb.x), __builtin_amdgcn_rsqf(nb.y)};
    }
    acos_ps_vn<NC>(cs, out);
}

__device__ __forceinline__ float angle3(f3 a, f3 b, f3 c) {
    f3 ba = sub3(a, b);
    f3 bc = sub3(c, b);
    return acos_ps((dot3_fast(ba, bc) * __builtin_amdgcn_rsqf(dot3_fast(ba, ba))) * __builtin_amdgcn_rsqf(dot3_fast(bc, bc)));
}

// ---- the FAITHFUL forms for NC columns x two rows (round 5): dihedral4_ref / angle3_ref bit for bit, packed ----
// The same operations as dihedral4_ref / angle3_ref (atan2_lib, acos_lib, the IEEE division), with the multiplies, adds and
// fused multiply-adds of two rows issued as one v_pk_*_f32 and the NC columns' dependent chains interleaved.  An IEEE fma /
// mul / add gives the same bits whether it issues as v_fma_f32 or as half of v_pk_fma_f32, so the results equal the scalar
// forms' bit for bit; that equality is not assumed but tested: tests/test_gpu_parity.py holds the sweep kernels (these
// forms) to the one-column kernel (the scalar forms) on every split, length and special value, and
// tools/microbench/libm_identity.hip sweeps 2^32 argument pairs through both and through the library calls.
//
// a / b, IEEE-correct: the expansion of fdiv (v_div_scale_f32 x2, v_rcp_f32, the Newton / residual chain, v_div_fmas_f32,
// v_div_fixup_f32); the chain's six operations packed.
template <int NC>
__device__ __forceinline__ void div_ieee_vn(const f32x2 (&a)[NC], const f32x2 (&b)[NC], f32x2 (&o)[NC]) {
    f32x2 ds[NC], ns[NC], r[NC], q[NC], t[NC];
    bool fx[NC], fy[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        bool unused;
        ds[c] = f32x2{__builtin_amdgcn_div_scalef(a[c].x, b[c].x, false, &unused), __builtin_amdgcn_div_scalef(a[c].y, b[c].y, false, &unused)};
        ns[c] = f32x2{__builtin_amdgcn_div_scalef(a[c].x, b[c].x, true, &fx[c]), __builtin_amdgcn_div_scalef(a[c].y, b[c].y, true, &fy[c])};
        r[c] = f32x2{__builtin_amdgcn_rcpf(ds[c].x), __builtin_amdgcn_rcpf(ds[c].y)};
    }
    const f32x2 one = {1.0f, 1.0f};
#pragma unroll
    for (int c = 0; c < NC; ++c) t[c] = __builtin_elementwise_fma(-ds[c], r[c], one);
#pragma unroll
    for (int c = 0; c < NC; ++c) r[c] = __builtin_elementwise_fma(t[c], r[c], r[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c) q[c] = ns[c] * r[c];
#pragma unroll
    for (int c = 0; c < NC; ++c) t[c] = __builtin_elementwise_fma(-ds[c], q[c], ns[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c) q[c] = __builtin_elementwise_fma(t[c], r[c], q[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c) t[c] = __builtin_elementwise_fma(-ds[c], q[c], ns[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
        o[c] = f32x2{__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(t[c].x, r[c].x, q[c].x, fx[c]), b[c].x, a[c].x),
                     __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(t[c].y, r[c].y, q[c].y, fy[c]), b[c].y, a[c].y)};
}

// atan2_lib on NC columns x two rows
template <int NC>
__device__ __forceinline__ void atan2_lib_vn(const f32x2 (&y)[NC], const f32x2 (&x)[NC], f32x2 (&o)[NC]) {
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 q[NC], t[NC], p[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float qq[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float ax = fabsf(h ? x[c].y : x[c].x), ay = fabsf(h ? y[c].y : y[c].x);
            const float v = __builtin_fminf(ax, ay), u = __builtin_fmaxf(ax, ay);
            const float m = __builtin_amdgcn_frexp_mantf(v) * __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(u));
            qq[h] = __builtin_amdgcn_ldexpf(m, __builtin_amdgcn_frexp_expf(v) - __builtin_amdgcn_frexp_expf(u));
        }
        q[c] = f32x2{qq[0], qq[1]};
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) t[c] = q[c] * q[c];
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(t[c], k2(__uint_as_float(0x3b2d2a58u)), k2(__uint_as_float(0xbc7a590cu)));
    constexpr uint32_t co[6] = {0x3d29fb3fu, 0xbd97d4d7u, 0x3dd931b2u, 0xbe1160e6u, 0x3e4cb8bfu, 0xbeaaaa62u};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(t[c], p[c], k2(__uint_as_float(co[s])));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = t[c] * p[c];
#pragma unroll
    for (int c = 0; c < NC; ++c) q[c] = __builtin_elementwise_fma(q[c], p[c], q[c]);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x2 t1 = k2(__uint_as_float(0x3fc90fdbu)) - q[c];
        f32x2 a = {fabsf(y[c].x) > fabsf(x[c].x) ? t1.x : q[c].x, fabsf(y[c].y) > fabsf(x[c].y) ? t1.y : q[c].y};
        const f32x2 t2 = k2(__uint_as_float(0x40490fdbu)) - a;
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float xe = h ? x[c].y : x[c].x, ye = h ? y[c].y : y[c].x;
            float e = (xe < 0.0f) ? (h ? t2.y : t2.x) : (h ? a.y : a.x);
            const float t3 = ((int)__float_as_uint(xe) < 0) ? __uint_as_float(0x40490fdbu) : 0.0f;
            e = (ye == 0.0f) ? t3 : e;
            const float t4 = (xe < 0.0f) ? __uint_as_float(0x4016cbe4u) : __uint_as_float(0x3f490fdbu);
            e = (__builtin_isinf(xe) && __builtin_isinf(ye)) ? t4 : e;
            e = (xe != xe || ye != ye) ? __uint_as_float(0x7fc00000u) : e;
            r[h] = copysignf(e, ye);
        }
        o[c] = f32x2{r[0], r[1]};
    }
}

// acos_lib on NC columns x two rows (the library's fma(c1, c2, -z) with c1 * c2 = pi, pi / 2 is, as hipcc folds it, a
// subtraction from the rounded constant)
template <int NC>
__device__ __forceinline__ void acos_lib_vn(const f32x2 (&x)[NC], f32x2 (&o)[NC]) {
    auto k2 = [](float c) { return f32x2{c, c}; };
    f32x2 w[NC], p[NC];
    bool big[NC][2];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x2 ax = {fabsf(x[c].x), fabsf(x[c].y)};
        const f32x2 h = __builtin_elementwise_fma(ax, k2(-0.5f), k2(0.5f)), s = x[c] * x[c];
        big[c][0] = ax.x > 0.5f; big[c][1] = ax.y > 0.5f;
        w[c] = f32x2{big[c][0] ? h.x : s.x, big[c][1] ? h.y : s.y};
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(w[c], k2(__uint_as_float(0x3d1c21a7u)), k2(__uint_as_float(0x3c5fc5dau)));
    constexpr uint32_t co[4] = {0x3d034c3cu, 0x3d3641b1u, 0x3d999bc8u, 0x3e2aaaacu};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) p[c] = __builtin_elementwise_fma(w[c], p[c], k2(__uint_as_float(co[s])));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x2 z = w[c] * p[c];
        const f32x2 sq = {__builtin_amdgcn_sqrtf(w[c].x), __builtin_amdgcn_sqrtf(w[c].y)};
        const f32x2 g = __builtin_elementwise_fma(sq, z, sq);
        const f32x2 g2 = g + g;
        const f32x2 neg = k2(__uint_as_float(0x40490fdbu)) - g2;
        const f32x2 sm = k2(__uint_as_float(0x3fc90fdbu)) - __builtin_elementwise_fma(x[c], z, x[c]);
        o[c] = f32x2{big[c][0] ? (x[c].x < 0.0f ? neg.x : g2.x) : sm.x, big[c][1] ? (x[c].y < 0.0f ? neg.y : g2.y) : sm.y};
    }
}

// dihedral4_ref for NC columns x two rows (the same operations per element in the same order)
template <int NC>
__device__ __forceinline__ void dihedral4v_ref_n(const f3v (&a)[NC], const f3v (&b)[NC], const f3v (&c)[NC],
                                                 const f3v (&d)[NC], f32x2 (&out)[NC]) {
    f32x2 x[NC], y0[NC], nb[NC], y[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const f3v b0 = sub3v(a[q], b[q]), b1 = sub3v(c[q], b[q]), b2 = sub3v(d[q], c[q]);
        const f3v n1 = cross3v(b0, b1), n2 = cross3v(b2, b1);
        const f3v m = cross3v(n1, n2);
        x[q] = dot3v_f(n1, n2);
        y0[q] = dot3v_f(m, b1);
        nb[q] = sqrt_rn_mk_v((b1.x * b1.x + b1.y * b1.y) + b1.z * b1.z);   // squares are never -0: dot3's leading `0 +` changes no bit
    }
    div_ieee_vn<NC>(y0, nb, y);
    atan2_lib_vn<NC>(y, x, out);
}

// angle3_ref for NC columns x two rows
template <int NC>
__device__ __forceinline__ void angle3v_ref_n(const f3v (&a)[NC], const f3v (&b)[NC], const f3v (&c)[NC], f32x2 (&out)[NC]) {
    f32x2 num[NC], den[NC], cs[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const f3v ba = sub3v(a[q], b[q]), bc = sub3v(c[q], b[q]);
        num[q] = dot3v_f(ba, bc);
        den[q] = sqrt_rn_mk_v((ba.x * ba.x + ba.y * ba.y) + ba.z * ba.z) * sqrt_rn_mk_v((bc.x * bc.x + bc.y * bc.y) + bc.z * bc.z);
    }
    div_ieee_vn<NC>(num, den, cs);
    acos_lib_vn<NC>(cs, out);
}

// geometry.gram_schmidt (geometry.py:428-439); e3 uses the last-axis cross
__device__ __forceinline__ void gram_schmidt3(f3 a, f3 b, f3 c, f3& e1, f3& e2, f3& e3) {
    f3 v1 = sub3(c, b);
    e1 = div3(v1, norm3(v1));
    f3 v2 = sub3(a, b);
    float p = dot3(e1, v2);
    f3 u2 = sub3(v2, scale3(e1, p));
    e2 = div3(u2, norm3(u2));
    e3 = cross3(e1, e2);
}

__device__ __forceinline__ f3 load3(const float* __restrict__ p) { return f3{p[0], p[1], p[2]}; }

// Launch `kernel` and return THIS launch's status.  hipLaunchKernel reports the result of the launch it performs;
// the sticky per-thread "last error" (which unrelated earlier HIP / PyTorch calls may have left set, and which
// hipGetLastError would both consume and misattribute) is neither read nor cleared.  Arguments are converted to
// the kernel's exact parameter types before their addresses are taken.
template <typename... KArgs, size_t... I>
static inline int ps_launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                 std::tuple<KArgs...>& params, std::index_sequence<I...>) {
    void* ptrs[] = {static_cast<void*>(&std::get<I>(params))...};
    return (int)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, stream);
}

template <typename... KArgs, typename... Args>
static inline int ps_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                            Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count does not match the kernel's parameter list");
    std::tuple<KArgs...> params(static_cast<KArgs>(args)...);
    return ps_launch_impl(kernel, grid, block, lds, stream, params, std::index_sequence_for<KArgs...>{});
}
