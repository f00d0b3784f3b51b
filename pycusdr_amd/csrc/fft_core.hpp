// fft_core.hpp -- register/LDS radix kernels for power-of-two inverse FFTs on gfx950.
//
// One "FFT group" of L/16 threads transforms L points.  Every thread keeps 16 complex points in
// registers; a pass of radix R runs 16/R in-register butterflies per thread and the passes are
// chained through one LDS exchange each.  Sign convention: inverse (+2*pi*i), unnormalised -- the
// convention of the matched-filter bank (cuFFT INVERSE, lib/cufft.py:129 in the reference).  A
// forward transform is obtained by conjugating input and output.
//
// Decomposition used by every pass (validated against numpy in tests/test_fft_algebra.py):
//   sub-FFT of length Lcur over index t, radix R, Lnext = Lcur/R
//   butterfly (prefix, t')  reads  pos = prefix*Lcur + t' + Lnext*i          i in [0,R)
//                           writes pos = (p*prefixCount + prefix)*Lnext + t' p in [0,R)
//                           after multiplying by W_Lcur^(t'*p)
//   after the last pass (Lnext == 1) the natural output index is  p*prefixCount + prefix.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// Complex values are 2-wide float vectors so that complex arithmetic maps onto the packed-fp32
// VALU ops of gfx950 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  Measured on MI355X
// (tools/ubench/valu.hip): one SIMD issues a wave64 fp32 VALU op every ~4.6 cycles, scalar OR packed,
// so a packed op does twice the work per issue slot; the FFT passes are VALU-issue-bound, which
// makes packing worth ~1.8x.  The swizzled forms (multiply by i, complex multiply) use the VOP3P
// op_sel / neg modifiers through inline asm: op_sel[i] / op_sel_hi[i] pick the half of source i
// that feeds the low / high result lane (defaults 0 / 1), neg_lo / neg_hi negate it.
typedef float cf __attribute__((ext_vector_type(2)));
#define DEVI __device__ __forceinline__

DEVI cf mkc(float x, float y) { return (cf){x, y}; }
DEVI cf cadd(cf a, cf b) { return a + b; }
DEVI cf csub(cf a, cf b) { return a - b; }
DEVI cf cconj(cf a) { return (cf){a.x, -a.y}; }
// a + i*b = (a.x - b.y, a.y + b.x)
DEVI cf add_i(cf a, cf b) {
    cf d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a - i*b = (a.x + b.y, a.y - b.x)
DEVI cf sub_i(cf a, cf b) {
    cf d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// complex product: (a.x*w.x - a.y*w.y, a.x*w.y + a.y*w.x) in two packed ops
// (one asm statement: the hazard recogniser pads every use of an asm-defined register that follows
// within one wait state with an s_nop, which between the two halves of a product is pure loss.  The
// first half goes straight into the destination: a separate temporary is the same physical register
// in two products that follow each other, and the recogniser then pads between THEM.)
DEVI cf cmul(cf a, cf w) {
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(d)
        : "v"(a), "v"(w));
    return d;
}
// conj(a) * w = (a.x*w.x + a.y*w.y, a.x*w.y - a.y*w.x): the forward transform of the segment search is
// run as conj(IFFT(conj(u))), and both conjugations fold into the multiplies next to it
DEVI cf cmul_cj(cf a, cf w) {
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
        : "=&v"(d)
        : "v"(a), "v"(w));
    return d;
}
// (x.x - x.y, x.x + x.y)  and  (-x.x - x.y, x.x - x.y)
DEVI cf rot45_sum(cf x) {
    cf d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(d) : "v"(x));
    return d;
}
DEVI cf rot135_sum(cf x) {
    cf d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[0,1]" : "=v"(d) : "v"(x));
    return d;
}

// ---- buffer (SRSRC) addressing: one VGPR byte offset + one SGPR offset per access, hardware range
// check (out-of-range loads return 0, stores are dropped).  Keeps 64-bit per-access address
// registers out of the unrolled 16-point load/store groups.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// Every descriptor in this library is wave-uniform by construction (per-lane parts go into the
// byte offset).  The compiler cannot always prove it -- anything derived from threadIdx, even
// threadIdx.x / 256 in a 256-thread workgroup, is divergent to it -- and would then wrap EACH buffer
// access in a "waterfall" loop (readfirstlane x4, compare, saveexec, access, loop).  Passing base and
// size through readfirstlane once per descriptor makes the uniformity explicit.
DEVI __amdgpu_buffer_rsrc_t mk_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(bytes);
    void *q = (void *)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, (int)nb, 0x00020000);
}
// Cache policy of the intermediate Z (written once by pass 1, read once by pass 2, 16 GiB apart):
// aux = 2 is the non-temporal hint ("nt").  Measured A/B at C2: nt on both the Z stores and the Z
// loads 6.93 -> 6.56 ms/block (keeps the XCD L2s for the filter tiles and the spectrum); nt on the
// loads alone is slower, sc0 (aux = 1) changes nothing.
#ifndef MFB_AUX_ZLOAD
#define MFB_AUX_ZLOAD 2
#endif
#ifndef MFB_AUX_ZSTORE
#define MFB_AUX_ZSTORE 2
#endif
template <int AUX = 0>
DEVI cf buf_load_cf(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    return mkc(__uint_as_float(v.x), __uint_as_float(v.y));
}
// two adjacent complex values with ONE 16-byte load: the texture addresser spends 16 cycles on a wave64
// buffer load whether a lane fetches 8 or 16 bytes (measured: TA_BUFFER_TOTAL_CYCLES / TA_BUFFER_WAVEFRONTS)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AUX = 0>
DEVI void buf_load_cf2(__amdgpu_buffer_rsrc_t r, int voff, int soff, cf &a, cf &b) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    a = mkc(__uint_as_float(v.x), __uint_as_float(v.y));
    b = mkc(__uint_as_float(v.z), __uint_as_float(v.w));
}
DEVI float buf_load_f(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <int AUX = 0>
DEVI void buf_store_cf(__amdgpu_buffer_rsrc_t r, int voff, int soff, cf val) {
    u32x2 v;
    v.x = __float_as_uint(val.x);
    v.y = __float_as_uint(val.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX);
}

constexpr int ilog2c(int x) { return x <= 1 ? 0 : 1 + ilog2c(x >> 1); }
// radix of pass s for a length-2^l transform: as many 16s as fit, then the remainder
constexpr int radix_of(int l, int s) { return (s < l / 4) ? 16 : ((s == l / 4 && (l % 4)) ? (1 << (l % 4)) : 0); }
constexpr int npass(int l) { return l / 4 + ((l % 4) ? 1 : 0); }
// register slot holding natural output p of an in-place radix-R butterfly
constexpr int rev(int R, int p) { return R == 16 ? 4 * (p % 4) + p / 4 : (R == 8 ? 2 * (p % 4) + p / 4 : p); }
// LDS padding: one extra element per 16 keeps stride-16 column reads off a single bank
DEVI int padi(int pos) { return pos + (pos >> 4); }
constexpr int padlen(int L) { return L + (L >> 4); }

template <int I, int N, class F>
DEVI void sfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// ---- in-register butterflies (inverse sign) ---------------------------------------------------
DEVI void b2(cf &a, cf &b) {
    cf t = a;
    a = t + b;
    b = t - b;
}
// (a, i*b) -> (a + i*b, a - i*b): radix-2 whose second input still has to be multiplied by i
DEVI void b2_bi(cf &a, cf &b) {
    cf t = a, u = b;
    a = add_i(t, u);
    b = sub_i(t, u);
}
DEVI void b4(cf &a0, cf &a1, cf &a2, cf &a3) {
    cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = add_i(t1, t3);  // t1 + i*t3
    a3 = sub_i(t1, t3);  // t1 - i*t3
}
// radix-4 whose third input (a2) still has to be multiplied by i
DEVI void b4_a2i(cf &a0, cf &a1, cf &a2, cf &a3) {
    cf t0 = add_i(a0, a2), t1 = sub_i(a0, a2), t2 = a1 + a3, t3 = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = add_i(t1, t3);
    a3 = sub_i(t1, t3);
}
#define MFB_H 0.70710678118654752440f
#define MFB_C 0.92387953251128675613f
#define MFB_S 0.38268343236508977173f
DEVI cf mul_w8_1(cf x) { return rot45_sum(x) * MFB_H; }   // e^{+i pi/4}
DEVI cf mul_w8_3(cf x) { return rot135_sum(x) * MFB_H; }  // e^{+i 3pi/4}
DEVI cf mul_w16_1(cf x) { return cmul(x, mkc(MFB_C, MFB_S)); }
DEVI cf mul_w16_3(cf x) { return cmul(x, mkc(MFB_S, MFB_C)); }
DEVI cf mul_w16_9(cf x) { return cmul(x, mkc(-MFB_C, -MFB_S)); }

template <int R, int O, int NV>
DEVI void bfly(cf (&v)[NV]) {
    if constexpr (R == 2) {
        b2(v[O], v[O + 1]);
    } else if constexpr (R == 4) {
        b4(v[O], v[O + 1], v[O + 2], v[O + 3]);
    } else if constexpr (R == 8) {
        b4(v[O + 0], v[O + 2], v[O + 4], v[O + 6]);
        b4(v[O + 1], v[O + 3], v[O + 5], v[O + 7]);
        v[O + 3] = mul_w8_1(v[O + 3]);
        v[O + 7] = mul_w8_3(v[O + 7]);
        b2(v[O + 0], v[O + 1]);
        b2(v[O + 2], v[O + 3]);
        b2_bi(v[O + 4], v[O + 5]);  // v[O+5] carries W8^2 = i, folded into the butterfly
        b2(v[O + 6], v[O + 7]);
    } else {
        static_assert(R == 16 && O == 0 && NV == 16, "radix");
        b4(v[0], v[4], v[8], v[12]);
        b4(v[1], v[5], v[9], v[13]);
        b4(v[2], v[6], v[10], v[14]);
        b4(v[3], v[7], v[11], v[15]);
        // u[t][p1] sits at t + 4*p1 ; twiddle W16^(t*p1)
        v[5] = mul_w16_1(v[5]);    // t=1,p1=1
        v[9] = mul_w8_1(v[9]);     // t=1,p1=2 -> W16^2
        v[13] = mul_w16_3(v[13]);  // t=1,p1=3
        v[6] = mul_w8_1(v[6]);     // t=2,p1=1 -> W16^2
        //  v[10]: t=2,p1=2 -> W16^4 = i, folded into the second-stage butterfly below
        v[14] = mul_w8_3(v[14]);   // t=2,p1=3 -> W16^6
        v[7] = mul_w16_3(v[7]);    // t=3,p1=1
        v[11] = mul_w8_3(v[11]);   // t=3,p1=2 -> W16^6
        v[15] = mul_w16_9(v[15]);  // t=3,p1=3 -> W16^9
        b4(v[0], v[1], v[2], v[3]);
        b4(v[4], v[5], v[6], v[7]);
        b4_a2i(v[8], v[9], v[10], v[11]);
        b4(v[12], v[13], v[14], v[15]);
    }
}


// ---- pass chain ----------------------------------------------------------------------------
// Twiddles.  Every pass but the last multiplies output p of butterfly (prefix, t') by
// W_Lcur^(t'*p).  Only radix-16 passes are ever followed by another pass, so a thread needs 15
// factors per twiddled pass, and they depend on the thread alone (not on the row / mask / Doppler
// bin being transformed).  TwRegs keeps them in registers across the work loop of a kernel
// (HOIST) -- or, for the very large transforms whose register budget is 128, reloads them from
// the table each time.
// LTW: only the first twiddled pass is kept in registers; the later ones (whose factors depend on the lane's position
// inside a group of L/256 lanes only) are read from a small table in LDS (broadcast reads), which frees 30 VGPRs per pass --
// what the barrier-team kernels need to run three waves per SIMD.
template <int L, bool LTW = false>
struct TwRegs {
    static constexpr int l = ilog2c(L);
    static constexpr int NTW = npass(l) - 1;  // twiddled passes
    static constexpr int NREG = LTW ? (NTW > 1 ? 1 : NTW) : NTW;      // of them in registers
    static constexpr bool LDS_TW = LTW;
    static constexpr int LSTRIDE = L / 256;   // lanes with distinct factors in pass 1 (L >= 512)
    static constexpr int LDS_ELEMS = LTW && NTW > 1 ? (NTW - 1) * 15 * (L / 256) : 0;
    cf r[NREG > 0 ? NREG : 1][16];
    const cf *ltab;                           // [pass - 1][p - 1][t]
};

template <int L, bool LTW, int S = 0>
DEVI void load_twiddles(TwRegs<L, LTW> &tw, const cf *__restrict__ table, const int g) {
    if constexpr (S < TwRegs<L, LTW>::NREG) {
        constexpr int Lcur = L >> (4 * S);
        constexpr int Lnext = Lcur / 16;
        constexpr int PC = L / Lcur;
        const int t = g % Lnext;  // NB == 1 for radix-16 passes: beta == g
        sfor<1, 16>([&](auto p) { tw.r[S][decltype(p)::value] = table[(t * decltype(p)::value) * PC]; });
        load_twiddles<L, LTW, S + 1>(tw, table, g);
    }
}
template <int L>
DEVI void load_twiddles(TwRegs<L, false> &tw, const cf *__restrict__ table, const int g) {
    load_twiddles<L, false, 0>(tw, table, g);
}
// the LDS part of an LTW set: `nthreads` threads fill ltab (the caller synchronises before the first transform)
template <int L>
DEVI void fill_lds_twiddles(cf *ltab, const cf *__restrict__ table, const int tid, const int nthreads) {
    using TW = TwRegs<L, true>;
    constexpr int l = ilog2c(L);
    for (int e = tid; e < TW::LDS_ELEMS; e += nthreads) {
        const int per = 15 * TW::LSTRIDE;
        const int S = 1 + e / per, rem = e % per;
        const int pp = 1 + rem / TW::LSTRIDE, t = rem % TW::LSTRIDE;
        const int Lcur = L >> (4 * S);
        const int Lnext = Lcur / 16 > 0 ? Lcur / 16 : 1;
        const int PC = L / Lcur;
        ltab[e] = table[((t % Lnext) * pp) * PC];
    }
    (void)l;
}

// Exchange synchronisation.  SYNC 0: the FFT group spans several wavefronts -> workgroup barrier.
// SYNC 1: the whole group (L/16 threads) lives inside ONE wavefront (L <= 1024): LDS operations of a
// wave execute in issue order, so a store followed by a load needs no s_barrier at all -- only the
// compiler must not reorder them.  Waves then run fully independently of each other.
template <int SYNC>
DEVI void xsync() {
    if constexpr (SYNC == 0) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- wave-local 2048-point transform: 32 points per lane, ONE exchange (seg_kernels.hpp, MFB_SEG_W32) --------------------
// 2048 = 32 (registers) x 64 (lanes).  Lane l stands at position g = pi(l) = (l >> 1) + 32 (l & 1); register i of the input
// holds element g + 64 i.
//   1. 32-point transform over i, in registers  ->  B[g][p]
//   2. twiddle W_2048^(g p) (31 per-lane factors, registers), row p of the exchange buffer <- B[g][p] (lane-contiguous stores)
//   3. lane l' = 2 p' + h reads row p' sixteen bytes at a time: (C[j][p'], C[32 + j][p']), j < 32 -- the values the lanes 2j and
//      2j + 1 wrote -- and combines them: h = 0: C1 + C2, h = 1: (C1 - C2) W_64^j  (the radix-2 level across the lane pair, done
//      on the data the exchange delivers anyway; its twiddle comes from a 2 x 32 table in LDS)
//   4. 32-point transform over j, in registers  ->  output m' of lane l' is element pi(l') + 64 m': the INPUT layout again.
// (tests/test_fft_algebra.py holds the numpy model of exactly this.)  One LDS round trip per transform instead of the two of
// the radix-16 chain, no cross-wave synchronisation at all.

// a * (S w), a * (S conj(w)), a * (S i w), a * (S i conj(w)) with S = +-1, w wave-uniform (SGPR pair: the twiddles inside the
// 32-point butterfly are a handful of constants; in VGPRs they would cost two registers each)
template <int CONJ, int ROT, int NEG>
DEVI cf cmul_k(cf a, cf w) {
    cf d;
    if constexpr (!CONJ && !ROT && !NEG)         // w
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(d) : "v"(a), "s"(w));
    else if constexpr (!CONJ && !ROT && NEG)     // -w = (-wx, -wy)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "=&v"(d) : "v"(a), "s"(w));
    else if constexpr (CONJ && !ROT && NEG)      // -conj(w) = (-wx, wy)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=&v"(d) : "v"(a), "s"(w));
    else if constexpr (!CONJ && ROT && !NEG)     // i w = (-wy, wx)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=&v"(d) : "v"(a), "s"(w));
    else if constexpr (CONJ && ROT && !NEG)      // i conj(w) = (wy, wx)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]" : "=&v"(d) : "v"(a), "s"(w));
    else {                                       // -i conj(w) = (-wy, -wx)
        static_assert(CONJ && ROT && NEG, "twiddle variant");
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]" : "=&v"(d) : "v"(a), "s"(w));
    }
    return d;
}
// i * x = (-x.y, x.x)
DEVI cf mul_i(cf x) {
    cf d;
    asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(x));
    return d;
}

#define MFB_C32_1X 0.98078528040323044913f   // cos(pi/16), sin(pi/16)
#define MFB_C32_1Y 0.19509032201612826785f
#define MFB_C32_3X 0.83146961230254523708f   // cos(3 pi/16), sin(3 pi/16)
#define MFB_C32_3Y 0.55557023301960222474f
// register slot that holds output p = p1 + 4 p2 of the in-register 32-point transform below
constexpr int slot32(int p) { return 8 * (p % 4) + rev(8, p / 4); }

// 32-point inverse-sign transform in registers: 8 radix-4 butterflies over (t, t+8, t+16, t+24), twiddles W_32^(t p1), 4 radix-8
// butterflies.  Output p sits in slot slot32(p).
DEVI void bfly32(cf (&v)[32]) {
    sfor<0, 8>([&](auto t) { constexpr int T = decltype(t)::value; b4(v[T], v[T + 8], v[T + 16], v[T + 24]); });
    const cf c1 = mkc(MFB_C32_1X, MFB_C32_1Y), c2 = mkc(MFB_C, MFB_S), c3 = mkc(MFB_C32_3X, MFB_C32_3Y);
    // slot t + 8 p1 carries W_32^(t p1)
    v[1 + 8] = cmul_k<0, 0, 0>(v[1 + 8], c1);     // 1
    v[1 + 16] = cmul_k<0, 0, 0>(v[1 + 16], c2);   // 2
    v[1 + 24] = cmul_k<0, 0, 0>(v[1 + 24], c3);   // 3
    v[2 + 8] = cmul_k<0, 0, 0>(v[2 + 8], c2);     // 2
    v[2 + 16] = mul_w8_1(v[2 + 16]);              // 4
    v[2 + 24] = cmul_k<1, 1, 0>(v[2 + 24], c2);   // 6 = i conj(W^2)
    v[3 + 8] = cmul_k<0, 0, 0>(v[3 + 8], c3);     // 3
    v[3 + 16] = cmul_k<1, 1, 0>(v[3 + 16], c2);   // 6
    v[3 + 24] = cmul_k<0, 1, 0>(v[3 + 24], c1);   // 9 = i W^1
    v[4 + 8] = mul_w8_1(v[4 + 8]);                // 4
    v[4 + 16] = mul_i(v[4 + 16]);                 // 8 = i
    v[4 + 24] = mul_w8_3(v[4 + 24]);              // 12
    v[5 + 8] = cmul_k<1, 1, 0>(v[5 + 8], c3);     // 5 = i conj(W^3)
    v[5 + 16] = cmul_k<0, 1, 0>(v[5 + 16], c2);   // 10 = i W^2
    v[5 + 24] = cmul_k<1, 0, 1>(v[5 + 24], c1);   // 15 = -conj(W^1)
    v[6 + 8] = cmul_k<1, 1, 0>(v[6 + 8], c2);     // 6
    v[6 + 16] = mul_w8_3(v[6 + 16]);              // 12
    v[6 + 24] = cmul_k<0, 0, 1>(v[6 + 24], c2);   // 18 = -W^2
    v[7 + 8] = cmul_k<1, 1, 0>(v[7 + 8], c1);     // 7 = i conj(W^1)
    v[7 + 16] = cmul_k<1, 0, 1>(v[7 + 16], c2);   // 14 = -conj(W^2)
    v[7 + 24] = cmul_k<1, 1, 1>(v[7 + 24], c3);   // 21 = -i conj(W^3)
    sfor<0, 4>([&](auto p1) { bfly<8, 8 * decltype(p1)::value>(v); });
}

struct W32Cfg {
    static constexpr int ROW = 66;                 // complex elements per row of the exchange buffer (64 + 2: conflict-free both ways)
    static constexpr int XCHG = 32 * ROW;          // per wave
    static constexpr int TW2_STRIDE = 34;          // [h][j]: 16-byte aligned rows on different banks (read two factors at a time)
    static constexpr int TW2_ELEMS = 2 * TW2_STRIDE;
};
struct W32Regs {
    cf tw1[32];          // W_2048^(pi(lane) p)
    cf sig;              // (+1, +1) on even lanes, (-1, -1) on odd ones
    const cf *tw2;       // this lane's row of the small table: [h][j] = h ? W_64^j : 1
};
// table: W_2048^j (inverse sign); ltw2: TW2_ELEMS elements of LDS, filled here by the whole workgroup (the caller synchronises)
DEVI void w32_setup(W32Regs &r, const cf *__restrict__ table, cf *ltw2, const int lane, const int tid, const int nthreads) {
    const int g = (lane >> 1) + 32 * (lane & 1);
    sfor<1, 32>([&](auto p) { r.tw1[decltype(p)::value] = table[(g * decltype(p)::value) & 2047]; });
    const float s = (lane & 1) ? -1.f : 1.f;
    r.sig = mkc(s, s);
    for (int e = tid; e < 64; e += nthreads) ltw2[(e >> 5) * W32Cfg::TW2_STRIDE + (e & 31)] = (e >> 5) ? table[32 * (e & 31)] : mkc(1.f, 0.f);
    r.tw2 = ltw2 + (lane & 1) * W32Cfg::TW2_STRIDE;
}
// v: slot i holds element pi(lane) + 64 i.  store(n, value, slot, nu) as in fft_passes: natural index n = nu + pi(lane), nu = 64 m'.
template <class Store>
DEVI void fft_w32(cf (&v)[32], cf *xbuf, const int lane, const W32Regs &r, Store &store) {
    bfly32(v);
    {
        cf *wr = xbuf + lane;
        sfor<0, 32>([&](auto p) {
            constexpr int P = decltype(p)::value;
            cf val = v[slot32(P)];
            if constexpr (P > 0) val = cmul(val, r.tw1[P]);
            wr[P * W32Cfg::ROW] = val;
        });
    }
    xsync<1>();
#ifndef MFB_W32_READS
#define MFB_W32_READS 3      // 1: 16-byte reads in the compiler's order; 2: two sweeps of 8-byte reads; 3: 16-byte reads, pipelined by hand
#endif
#ifndef MFB_W32_DEPTH
#define MFB_W32_DEPTH 3
#endif
#if MFB_W32_READS == 1
    {
        // (16-byte reads throughout: ds_read_b128 moves 16 bytes per lane in 4 LDS cycles, ds_read2_b64 -- what the compiler
        // makes of two adjacent 8-byte reads -- in 8)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *rd = reinterpret_cast<const f4 *>(xbuf + (lane >> 1) * W32Cfg::ROW);
        const f4 *tw = reinterpret_cast<const f4 *>(r.tw2);
        sfor<0, 16>([&](auto jj) {
            constexpr int J = 2 * decltype(jj)::value;
            const f4 q0 = rd[J], q1 = rd[J + 1], f = tw[J / 2];
            cf t0 = __builtin_elementwise_fma(mkc(q0.z, q0.w), r.sig, mkc(q0.x, q0.y));
            cf t1 = __builtin_elementwise_fma(mkc(q1.z, q1.w), r.sig, mkc(q1.x, q1.y));
            if constexpr (J > 0) t0 = cmul(t0, mkc(f.x, f.y));
            t1 = cmul(t1, mkc(f.z, f.w));
            v[J] = t0;
            v[J + 1] = t1;
        });
    }
#elif MFB_W32_READS == 3
    {
        // The same 16-byte reads, software-pipelined by hand: the reads of pair k + DEPTH are issued before pair k is combined,
        // and the order is pinned with scheduling barriers.  (Left to itself the scheduler issues the two reads of a pair and
        // waits for them at once -- the register pressure of the region puts it in its pressure-first mode -- so every pair
        // exposed a full LDS round trip: ~16 x 100+ cycles per transform that only the other wave of the SIMD could cover.)
        // Pairs go out in the order the second butterfly's first radix-4 level consumes them (j, j + 8, j + 16, j + 24).
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *rd = reinterpret_cast<const f4 *>(xbuf + (lane >> 1) * W32Cfg::ROW);
        const f4 *tw = reinterpret_cast<const f4 *>(r.tw2);
        constexpr int DEPTH = MFB_W32_DEPTH;
        f4 q0[16], q1[16], f[16];
        auto issue = [&](auto kk) {
            constexpr int K = decltype(kk)::value;
            constexpr int J = 2 * (K >> 2) + 8 * (K & 3);           // k-th pair issued -> elements J, J + 1
            q0[K] = rd[J];
            q1[K] = rd[J + 1];
            f[K] = tw[J / 2];
        };
        sfor<0, DEPTH>([&](auto kk) { issue(kk); });
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, 16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            constexpr int J = 2 * (K >> 2) + 8 * (K & 3);
            if constexpr (K + DEPTH < 16) {
                issue(std::integral_constant<int, K + DEPTH>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            cf t0 = __builtin_elementwise_fma(mkc(q0[K].z, q0[K].w), r.sig, mkc(q0[K].x, q0[K].y));
            cf t1 = __builtin_elementwise_fma(mkc(q1[K].z, q1[K].w), r.sig, mkc(q1[K].x, q1[K].y));
            if constexpr (J > 0) t0 = cmul(t0, mkc(f[K].x, f[K].y));
            t1 = cmul(t1, mkc(f[K].z, f[K].w));
            v[J] = t0;
            v[J + 1] = t1;
            __builtin_amdgcn_sched_barrier(0);
        });
    }
#else
    {
        // two sweeps of 8-byte reads: all 32 first halves go straight into v (every read in flight at once: no temporaries),
        // then the second halves are folded in as they arrive.  (One 16-byte read per point needs four registers per read in
        // flight beside v: the compiler kept three or four in flight and the wave sat in s_waitcnt.)
        const cf *rd = xbuf + (lane >> 1) * W32Cfg::ROW;
        sfor<0, 32>([&](auto j) { v[decltype(j)::value] = rd[2 * decltype(j)::value]; });
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, 32>([&](auto j) {
            constexpr int J = decltype(j)::value;
            cf t = __builtin_elementwise_fma(rd[2 * J + 1], r.sig, v[J]);
            if constexpr (J > 0) t = cmul(t, r.tw2[J]);
            v[J] = t;
        });
    }
#endif
    xsync<1>();      // the rows are rewritten by the next transform of this wave
    bfly32(v);
    sfor<0, 32>([&](auto m) {
        constexpr int M = decltype(m)::value;
        store(64 * M + ((lane >> 1) + 32 * (lane & 1)), v[slot32(M)], std::integral_constant<int, M>{}, std::integral_constant<int, 64 * M>{});
    });
}

// ---- fused-twiddle radix-2 transforms (round 6; MFB_FFT_FUSED) ------------------------------------------------------------
// The radix-16 chain above multiplies by every twiddle on its own -- v_pk_mul + v_pk_fma -- and adds afterwards: a radix-2 butterfly
// a +- w b costs four packed operations.  With the twiddle in "tangent form", w = c (1 + i t), t = tan(arg w), c = cos(arg w)
// (Linzer & Feig's fused-multiply-add butterflies), it costs THREE, all v_pk_fma_f32:
//       b' = b + t (i b)          a + c b'          a - c b'
// and one rounding less per output.  A transform is then a radix-2 decimation-in-time chain in which only ONE input of a butterfly
// ever carries a twiddle.  For the second pass of a two-pass transform this needs the inter-pass twiddle on the READ side of
// the exchange: lane p reads b_i = B[i][p], i < n, and owes out[q] = sum_i b_i (W_n^q w)^i with w = W_L^p, which is the recursion
//       F(b, w, n)[q], F[q + n/2] = F(b_even, w^2, n/2)[q] +- (W_n^q w) F(b_odd, w^2, n/2)[q]
// -- stage s (sub-transforms of 2^s points) uses the twiddles W_(2^s)^q w^(n >> s), q < 2^(s-1); q and q + 2^(s-2) differ by the
// factor i, which is an operand swizzle (ROT), so a lane keeps 1 + 1 + 2 + 4 (+ 8) = 8 (16) pairs (c, t) for n = 16 (32) -- 16
// (32) VGPRs where the radix-16 chain kept 30 (62) -- and none of their angles reaches pi/2 except in stage 1, where the one lane
// whose twiddle is exactly i gets (2^-40, 2^40): c t = 1 exactly and c b vanishes in the rounding.  In-place slots: stage s pairs
// slot a with a + (n >> s); output q ends in slot bitrev(q).  tests/test_fft_algebra.py holds the numpy model (fp32 rounding per
// operation) of exactly this; per 256-point transform 170 packed operations instead of 190, per 2048-point transform 498
// instead of 655 (profiles/r06_fft_ops.md).
#ifndef MFB_FFT_FUSED
#define MFB_FFT_FUSED 1
#endif
// the write side (constant twiddles) as a fused chain too (1), or as the radix-4 butterflies of the older transforms (0: fewer
// multiplier operations, more instructions -- the A/B of profiles/r06_fft_ops.md)
#ifndef MFB_FUSED_FIRST
#define MFB_FUSED_FIRST 1
#endif
constexpr int brevc(int x, int bits) { return bits <= 0 ? 0 : (((x & 1) << (bits - 1)) | brevc(x >> 1, bits - 1)); }

// (a, b) <- (a + w b, a - w b), w = i^ROT c (1 + i t), tw = (c, t) in a VGPR pair (SREG: an SGPR pair).  NX / NY: which of the
// two outputs is wanted (a dead half of a last-stage butterfly is not computed).  One asm statement per butterfly: the hazard
// recogniser would pad every hand-over between separate statements (see cmul).  b' overwrites b, the sum goes to a fresh pair.
#define MFB_I1(B) "v_pk_fma_f32 %[" B "], %[w], %[" B "], %[" B "] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
#define MFB_IX0(X, B, A) "v_pk_fma_f32 %[" X "], %[w], %[" B "], %[" A "] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
#define MFB_IY0(B, A) "v_pk_fma_f32 %[" B "], %[w], %[" B "], %[" A "] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
#define MFB_IX1(X, B, A) "v_pk_fma_f32 %[" X "], %[w], %[" B "], %[" A "] op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]\n\t"
#define MFB_IY1(B, A) "v_pk_fma_f32 %[" B "], %[w], %[" B "], %[" A "] op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]\n\t"
template <int ROT, bool NX, bool NY, bool SREG>
DEVI void bf_ct(cf &a, cf &b, const cf tw) {
    static_assert(NX || NY, "a butterfly nobody wants");
    cf x;
#define MFB_BF1(BODY, OUTS)                                                          \
    do {                                                                             \
        if constexpr (SREG) asm(BODY : OUTS : [w] "s"(tw), [a] "v"(a));              \
        else asm(BODY : OUTS : [w] "v"(tw), [a] "v"(a));                             \
    } while (0)
#define MFB_OUT_XB [x] "=&v"(x), [b] "+v"(b)
#define MFB_OUT_B [b] "+v"(b)
    if constexpr (NX && NY) {
        if constexpr (ROT) MFB_BF1(MFB_I1("b") MFB_IX1("x", "b", "a") MFB_IY1("b", "a"), MFB_OUT_XB);
        else MFB_BF1(MFB_I1("b") MFB_IX0("x", "b", "a") MFB_IY0("b", "a"), MFB_OUT_XB);
        a = x;
    } else if constexpr (NX) {
        if constexpr (ROT) MFB_BF1(MFB_I1("b") MFB_IX1("x", "b", "a"), MFB_OUT_XB);
        else MFB_BF1(MFB_I1("b") MFB_IX0("x", "b", "a"), MFB_OUT_XB);
        a = x;
    } else {
        if constexpr (ROT) MFB_BF1(MFB_I1("b") MFB_IY1("b", "a"), MFB_OUT_B);
        else MFB_BF1(MFB_I1("b") MFB_IY0("b", "a"), MFB_OUT_B);
    }
#undef MFB_BF1
#undef MFB_OUT_XB
#undef MFB_OUT_B
}
// Two butterflies with the SAME twiddle pair in one statement, instruction by instruction in turn: inside one butterfly every
// instruction waits for the one before it, and the compiler cannot move anything into an asm statement -- two side by side give
// the wave an independent instruction between each dependent pair.  ROTA / ROTB as above; NYB = false: the second butterfly's
// difference is dead (last stage of a transform whose upper outputs nobody wants).
// Measured (profiles/r06_fft_ops.md, one device, three banks): pairs are 0 ... 1 % SLOWER than single butterflies -- with three
// (two) waves per SIMD the other waves fill the slot behind a dependent instruction anyway, and a six-instruction statement takes
// freedom from the scheduler.  Off; kept as the knob of that A/B.
#ifndef MFB_BF_PAIR
#define MFB_BF_PAIR 0
#endif
template <int ROTA, int ROTB, bool NYB, bool SREG>
DEVI void bf_ct2(cf &a, cf &b, cf &c, cf &d, const cf tw) {
    cf x, y;
#define MFB_BF2(BODY)                                                                                                            \
    do {                                                                                                                         \
        if constexpr (SREG) asm(BODY : [x] "=&v"(x), [b] "+v"(b), [y] "=&v"(y), [d] "+v"(d) : [w] "s"(tw), [a] "v"(a), [c] "v"(c)); \
        else asm(BODY : [x] "=&v"(x), [b] "+v"(b), [y] "=&v"(y), [d] "+v"(d) : [w] "v"(tw), [a] "v"(a), [c] "v"(c));               \
    } while (0)
    if constexpr (!ROTA && !ROTB) {
        static_assert(NYB, "same-rotation pairs are whole");
        MFB_BF2(MFB_I1("b") MFB_I1("d") MFB_IX0("x", "b", "a") MFB_IX0("y", "d", "c") MFB_IY0("b", "a") MFB_IY0("d", "c"));
    } else if constexpr (ROTA && ROTB) {
        static_assert(NYB, "same-rotation pairs are whole");
        MFB_BF2(MFB_I1("b") MFB_I1("d") MFB_IX1("x", "b", "a") MFB_IX1("y", "d", "c") MFB_IY1("b", "a") MFB_IY1("d", "c"));
    } else {
        static_assert(!ROTA && ROTB, "last-stage pairs: q and q + half");
        if constexpr (NYB) MFB_BF2(MFB_I1("b") MFB_I1("d") MFB_IX0("x", "b", "a") MFB_IX1("y", "d", "c") MFB_IY0("b", "a") MFB_IY1("d", "c"));
        else MFB_BF2(MFB_I1("b") MFB_I1("d") MFB_IX0("x", "b", "a") MFB_IX1("y", "d", "c") MFB_IY0("b", "a"));
    }
#undef MFB_BF2
    a = x;
    c = y;
}
// e = a + c (b + t (i b)): half a butterfly (the lane pair of the 2048-point transform shares one: each lane keeps its own sign in
// c), two of them side by side
DEVI void half_ct2(cf &e0, cf &e1, const cf a0, cf b0, const cf a1, cf b1, const cf tw) {
    asm(MFB_I1("b") MFB_I1("d") MFB_IX0("x", "b", "a") MFB_IX0("y", "d", "c")
        : [x] "=&v"(e0), [b] "+v"(b0), [y] "=&v"(e1), [d] "+v"(b1)
        : [w] "v"(tw), [a] "v"(a0), [c] "v"(a1));
}
// (cos, tan) of k pi / 16: the constant twiddles inside a 16- or 32-point transform
template <int K> struct CtK;
template <> struct CtK<1> { static constexpr float c = 0.98078528040323043058f, t = 0.19891236737965800607f; };
template <> struct CtK<2> { static constexpr float c = 0.92387953251128673848f, t = 0.41421356237309503445f; };
template <> struct CtK<3> { static constexpr float c = 0.83146961230254523567f, t = 0.66817863791929887896f; };
template <> struct CtK<4> { static constexpr float c = 0.70710678118654757274f, t = 1.0f; };
template <> struct CtK<5> { static constexpr float c = 0.55557023301960228867f, t = 1.49660576266548894786f; };
template <> struct CtK<6> { static constexpr float c = 0.38268343236508983729f, t = 2.41421356237309492343f; };
template <> struct CtK<7> { static constexpr float c = 0.19509032201612833135f, t = 5.02733949212584629862f; };

// table index of the twiddle pair of stage s (1-based), reduced index qb
constexpr int ct_half(int s) { return s >= 2 ? (1 << (s >= 2 ? s - 2 : 0)) : 1; }        // twiddles of stage s up to the factor i
constexpr int ct_index(int s, int qb) { return (s == 1 ? 0 : ct_half(s)) + qb; }
constexpr int ct_count(int levels) { return 1 << (levels - 1); }

// In-place radix-2 decimation-in-time transform of n = 2^LEVELS register slots (natural input order; output q in slot
// brevc(q, LEVELS)), stages S0 ... LEVELS.  LANE_TW: per-lane twiddles twr[ct_index(s, qb)] (the second pass of a two-pass
// transform, above); otherwise the plain n-point transform with its constant twiddles (from SGPR pairs).  Only outputs
// q < LIVE are wanted.
// slot of the u-th butterfly's first input in a stage that pairs a with a + off: u with a zero bit inserted at off
constexpr int bf_slot(int u, int off) { return (u / off) * 2 * off + (u % off); }
// Is the value that slot `slot` holds AFTER stage s wanted, when only the outputs q in [lo, hi) are?  After the last stage the slot
// holds output brevc(slot); a value of an earlier stage is wanted when one of the two outputs of its next butterfly is.
constexpr bool dit_need(int levels, int s, int slot, int lo, int hi) {
    if (s >= levels) return brevc(slot, levels) >= lo && brevc(slot, levels) < hi;
    const int off = (1 << levels) >> (s + 1);
    return dit_need(levels, s + 1, slot, lo, hi) || dit_need(levels, s + 1, slot ^ off, lo, hi);
}
// Only the outputs q in [LO, HI) are wanted: every butterfly half that feeds none of them is left out (round 6: LO > 0 is the
// "complement" form of the search -- the energy of the INVALID outputs of a segment, seg_kernels.hpp segf_body).
template <int LEVELS, int LO, int HI, bool LANE_TW, int S0 = 1, int NTW>
DEVI void dit_fused(cf (&v)[1 << LEVELS], const cf (&twr)[NTW]) {
    constexpr int n = 1 << LEVELS;
    sfor<S0, LEVELS + 1>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        constexpr int off = n >> s;
        constexpr int half = ct_half(s);
        // butterflies u = 2k, 2k + 1 share their twiddle pair: the same q in every stage but the last, q and q + half -- a
        // rotation by i -- in the last
        sfor<0, n / 4>([&](auto k_) {
            constexpr int ua = 2 * decltype(k_)::value, ub = ua + 1;
            constexpr int a = bf_slot(ua, off), c = bf_slot(ub, off);
            constexpr int qa = brevc(a >> (LEVELS - s + 1), s - 1), qc = brevc(c >> (LEVELS - s + 1), s - 1);
            constexpr bool rota = s >= 2 && qa >= half, rotc = s >= 2 && qc >= half;
            constexpr int qb = rota ? qa - half : qa;
            static_assert(qb == (rotc ? qc - half : qc), "a pair shares its twiddle");
            constexpr bool nxa = dit_need(LEVELS, s, a, LO, HI), nya = dit_need(LEVELS, s, a + off, LO, HI);
            constexpr bool nxc = dit_need(LEVELS, s, c, LO, HI), nyc = dit_need(LEVELS, s, c + off, LO, HI);
            auto one = [&](auto slot_, auto rot_, auto nx_, auto ny_) {
                constexpr int sl = decltype(slot_)::value;
                constexpr bool rot = decltype(rot_)::value, nx = decltype(nx_)::value, ny = decltype(ny_)::value;
                if constexpr (nx || ny) {
                    if constexpr (LANE_TW) {
                        bf_ct<rot, nx, ny, false>(v[sl], v[sl + off], twr[ct_index(s, qb)]);
                    } else if constexpr (qb == 0) {
                        if constexpr (rot) b2_bi(v[sl], v[sl + off]);
                        else b2(v[sl], v[sl + off]);
                    } else {
                        using K = CtK<qb * (32 >> s)>;
                        bf_ct<rot, nx, ny, true>(v[sl], v[sl + off], mkc(K::c, K::t));
                    }
                }
            };
            constexpr bool pairable = MFB_BF_PAIR && (LANE_TW || qb != 0) && nxa && nya && nxc && (nyc || (!rota && rotc)) &&
                                      (rota == rotc || (!rota && rotc));
            if constexpr (pairable) {
                if constexpr (LANE_TW) {
                    static_assert(NTW >= ct_count(LEVELS), "twiddle pairs");
                    bf_ct2<rota, rotc, nyc, false>(v[a], v[a + off], v[c], v[c + off], twr[ct_index(s, qb)]);
                } else {
                    static_assert(s <= 5, "constant twiddles of up to 32 points");
                    using K = CtK<qb * (32 >> s)>;
                    bf_ct2<rota, rotc, nyc, true>(v[a], v[a + off], v[c], v[c + off], mkc(K::c, K::t));
                }
            } else {
                one(std::integral_constant<int, a>{}, std::integral_constant<bool, rota>{}, std::integral_constant<bool, nxa>{}, std::integral_constant<bool, nya>{});
                one(std::integral_constant<int, c>{}, std::integral_constant<bool, rotc>{}, std::integral_constant<bool, nxc>{}, std::integral_constant<bool, nyc>{});
            }
        });
    });
}

// 256 points = 16 lanes x 16 registers, two fused 16-point transforms around one exchange (exchange image as in fft_passes:
// element pos at padi(pos)).  Lane g: register i holds element g + 16 i on entry; output q is element g + 16 q.
//   ct: the lane's 8 twiddle pairs, (cos, tan) of 2 pi u / 256 with u = 8g | 4g | 2g, 2g + 32 | g, g + 16, g + 32, g + 48
struct F256Regs {
    cf ct[8];
};
// table: [16 lanes][8] pairs behind the W_L table (mfbank.hip, append_fused_tables)
DEVI void f256_setup(F256Regs &r, const cf *__restrict__ table, const int g) {
    sfor<0, 8>([&](auto k) { r.ct[decltype(k)::value] = table[g * 8 + decltype(k)::value]; });
}
template <int LO, int HI, class Store>
DEVI void fft256_fused(cf (&v)[16], cf *buf, const int g, const F256Regs &r, Store &store) {
#if MFB_FUSED_FIRST
    const cf none[1] = {mkc(0.f, 0.f)};
    dit_fused<4, 0, 16, false>(v, none);
    {
        cf *wr = buf + g;
        sfor<0, 16>([&](auto p) { wr[decltype(p)::value * 17] = v[brevc(decltype(p)::value, 4)]; });      // padi(16 p + g)
    }
#else
    bfly<16, 0>(v);
    {
        cf *wr = buf + g;
        sfor<0, 16>([&](auto p) { wr[decltype(p)::value * 17] = v[rev(16, decltype(p)::value)]; });
    }
#endif
    xsync<1>();
    {
        const cf *rd = buf + g * 17;                                                                      // padi(16 g + i)
        sfor<0, 16>([&](auto i) { v[decltype(i)::value] = rd[decltype(i)::value]; });
    }
    xsync<1>();
    dit_fused<4, LO, HI, true>(v, r.ct);
    sfor<0, 16>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        store(16 * Q + g, v[brevc(Q, 4)], std::integral_constant<int, Q>{}, std::integral_constant<int, 16 * Q>{});
    });
}

// 2048 points = 64 lanes x 32 registers (layout of fft_w32 above: lane l at pi(l) = (l >> 1) + 32 (l & 1), register i = element
// pi(l) + 64 i, output m' = element pi(l) + 64 m').  Write side: plain 32-point transform, rows of the exchange buffer untwiddled.
// Read side, lane l' = 2 p' + h: e_j = b_j + (-1)^h W_64^p' b_(j+32) -- HALF a fused butterfly, the sign in the lane's c -- then
// F(e, W_2048^pi(l'), 32): the inter-pass twiddle, the lane-pair level's twiddle W_64^(h j) and the second transform's own are one set
// of 16 pairs per lane.
struct F2048Regs {
    cf ct[16];     // (cos, tan) of 2 pi u / 2048, u = 16 g' | 8 g' | 4 g' + 256 q (q < 2) | 2 g' + 128 q (q < 4) | g' + 64 q (q < 8), g' = pi(lane)
    cf t0;         // (+-cos, tan) of 2 pi p' / 64, p' = lane >> 1; minus on odd lanes
};
// table: [64 positions][16] pairs, then [32] pairs, behind the W_L table
DEVI void f2048_setup(F2048Regs &r, const cf *__restrict__ table, const int lane) {
    const int g = (lane >> 1) + 32 * (lane & 1);
    sfor<0, 16>([&](auto k) { r.ct[decltype(k)::value] = table[g * 16 + decltype(k)::value]; });
    cf t0 = table[64 * 16 + (lane >> 1)];
    if (lane & 1) t0.x = -t0.x;
    r.t0 = t0;
}
#ifndef MFB_F2048_DEPTH
#define MFB_F2048_DEPTH 4
#endif
template <int LO, int HI, class Store>
DEVI void fft_w32_fused(cf (&v)[32], cf *xbuf, const int lane, const F2048Regs &r, Store &store) {
#if MFB_FUSED_FIRST
    const cf none[1] = {mkc(0.f, 0.f)};
    dit_fused<5, 0, 32, false>(v, none);
    {
        cf *wr = xbuf + lane;
        sfor<0, 32>([&](auto p) { wr[decltype(p)::value * W32Cfg::ROW] = v[brevc(decltype(p)::value, 5)]; });
    }
#else
    bfly32(v);
    {
        cf *wr = xbuf + lane;
        sfor<0, 32>([&](auto p) { wr[decltype(p)::value * W32Cfg::ROW] = v[slot32(decltype(p)::value)]; });
    }
#endif
    xsync<1>();
    {
        // 16-byte reads (elements 2J, 2J + 1 of the row = what lanes 2J and 2J + 1 wrote = positions J and J + 32), software-
        // pipelined by hand as in fft_w32; pairs go out in the order stage 1 of the second transform pairs them (j, j + 16), and
        // its butterflies run inside the loop, as soon as both halves are in
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *rd = reinterpret_cast<const f4 *>(xbuf + (lane >> 1) * W32Cfg::ROW);
        constexpr int DEPTH = MFB_F2048_DEPTH;
        f4 q0[16], q1[16];
        auto issue = [&](auto kk) {
            constexpr int K = decltype(kk)::value;
            constexpr int J = 2 * (K >> 1) + 16 * (K & 1);          // k-th pair issued -> elements J, J + 1
            q0[K] = rd[J];
            q1[K] = rd[J + 1];
        };
        sfor<0, DEPTH>([&](auto kk) { issue(kk); });
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, 16>([&](auto kk) {
            constexpr int K = decltype(kk)::value;
            constexpr int J = 2 * (K >> 1) + 16 * (K & 1);
            if constexpr (K + DEPTH < 16) {
                issue(std::integral_constant<int, K + DEPTH>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            half_ct2(v[J], v[J + 1], mkc(q0[K].x, q0[K].y), mkc(q0[K].z, q0[K].w), mkc(q1[K].x, q1[K].y), mkc(q1[K].z, q1[K].w), r.t0);
            if constexpr (K & 1) bf_ct2<0, 0, true, false>(v[J - 16], v[J], v[J - 15], v[J + 1], r.ct[0]);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    xsync<1>();      // the rows are rewritten by the next transform of this wave
    dit_fused<5, LO, HI, true, 2>(v, r.ct);
    sfor<0, 32>([&](auto m) {
        constexpr int M = decltype(m)::value;
        store(64 * M + ((lane >> 1) + 32 * (lane & 1)), v[brevc(M, 5)], std::integral_constant<int, M>{}, std::integral_constant<int, 64 * M>{});
    });
}

// v    : the thread's 16 points; on entry of pass 0 slot i holds element g + (L/16)*i
// lds  : exchange buffer: element (pos, col) at padi(pos)*T + col.  With PP (ping-pong) the buffer
//        holds two halves of HALF elements used alternately (one barrier per exchange: a half is
//        rewritten only after a later barrier that every reader of it has passed); without PP one
//        half is used and every exchange takes two barriers.
// ebuf : which half the next exchange uses (caller keeps it across calls)
// g    : thread index inside the FFT group, 0 <= g < L/16 ;  col: column inside the tile
// store(n, value, slot, nu): natural output index n = nu + g; slot (register slot id) and nu are
//        compile-time constants (std::integral_constant)
template <int L, int T, int S, bool HOIST, bool PP, int HALF, int SYNC = 0, class TW, class Store>
DEVI void fft_passes(cf (&v)[16], cf *lds, int &ebuf, const int g, const int col, const TW &twr,
                     const cf *__restrict__ table, Store &store) {
    constexpr int l = ilog2c(L);
    constexpr int NP = npass(l);
    constexpr int R = radix_of(l, S);
    constexpr int Lcur = L >> (4 * S);
    constexpr int Lnext = Lcur / R;
    constexpr int PC = L / Lcur;  // prefixCount
    constexpr int NT = L / 16;
    constexpr int NB = 16 / R;

    sfor<0, NB>([&](auto u) { bfly<R, decltype(u)::value * R>(v); });

    if constexpr (S == NP - 1) {
        sfor<0, NB>([&](auto u) {
            sfor<0, R>([&](auto p) {
                constexpr int slot = decltype(u)::value * R + decltype(p)::value;
                // natural index n = nu + g with nu a compile-time constant (wave-uniform part)
                constexpr int nu = decltype(p)::value * PC + NT * decltype(u)::value;
                store(nu + g, v[decltype(u)::value * R + rev(R, decltype(p)::value)], std::integral_constant<int, slot>{},
                      std::integral_constant<int, nu>{});
            });
        });
    } else {
        static_assert(R == 16 && NB == 1, "only radix-16 passes are followed by another pass");
        cf *buf = lds;
        if constexpr (PP) {
            buf = lds + (ebuf ? HALF : 0);
            ebuf ^= 1;
        }
        {
            const int prefix = g / Lnext;
            const int t = g % Lnext;
            sfor<0, R>([&](auto p) {
                constexpr int pp = decltype(p)::value;
                cf val = v[rev(R, pp)];
                if constexpr (pp > 0) {
                    if constexpr (HOIST && TW::LDS_TW && S >= TW::NREG)
                        val = cmul(val, twr.ltab[((S - 1) * 15 + (pp - 1)) * TW::LSTRIDE + t % TW::LSTRIDE]);
                    else if constexpr (HOIST) val = cmul(val, twr.r[S][pp]);
                    else val = cmul(val, table[(t * pp) * PC]);
                }
                buf[padi((pp * PC + prefix) * Lnext + t) * T + col] = val;
            });
        }
        xsync<SYNC>();
        constexpr int R2 = radix_of(l, S + 1);
        constexpr int Lnn = Lnext / R2;
        constexpr int NB2 = 16 / R2;
        sfor<0, NB2>([&](auto u) {
            const int beta = g + NT * decltype(u)::value;
            const int prefix = beta / Lnn;
            const int t = beta % Lnn;
            sfor<0, R2>([&](auto i) {
                v[decltype(u)::value * R2 + decltype(i)::value] =
                    buf[padi(prefix * Lnext + t + Lnn * decltype(i)::value) * T + col];
            });
        });
        if constexpr (!PP) xsync<SYNC>();
        fft_passes<L, T, S + 1, HOIST, PP, HALF, SYNC>(v, lds, ebuf, g, col, twr, table, store);
    }
}

// Two transforms of the same team in lock-step, one exchange buffer each (lds, lds + HALF).  Transform B's butterflies sit
// between A's exchange stores and loads, and after an exchange both sets of loads are in flight together while A's next
// butterflies already run: a wave covers its own exchange latency instead of relying on the other wave of its SIMD.
// Barriers: one before a level's stores (every wave is done reading both buffers), one between stores and loads: two per
// level for two transforms, what alternating buffers give a single transform.  The caller must put a barrier in front of
// whatever uses the buffers next.
template <int L, int T, int S, bool HOIST, int HALF, int SYNC, class TW, class StoreA, class StoreB>
DEVI void fft_passes2(cf (&va)[16], cf (&vb)[16], cf *lds, const int g, const int col, const TW &twr, const cf *__restrict__ table,
                      StoreA &sa, StoreB &sb) {
    constexpr int l = ilog2c(L);
    constexpr int NP = npass(l);
    constexpr int R = radix_of(l, S);
    constexpr int Lcur = L >> (4 * S);
    constexpr int Lnext = Lcur / R;
    constexpr int PC = L / Lcur;
    constexpr int NT = L / 16;
    constexpr int NB = 16 / R;

    sfor<0, NB>([&](auto u) { bfly<R, decltype(u)::value * R>(va); });
    if constexpr (S == NP - 1) {
        sfor<0, NB>([&](auto u) { bfly<R, decltype(u)::value * R>(vb); });
        auto emit = [&](cf (&v)[16], auto &store) {
            sfor<0, NB>([&](auto u) {
                sfor<0, R>([&](auto p) {
                    constexpr int slot = decltype(u)::value * R + decltype(p)::value;
                    constexpr int nu = decltype(p)::value * PC + NT * decltype(u)::value;
                    store(nu + g, v[decltype(u)::value * R + rev(R, decltype(p)::value)], std::integral_constant<int, slot>{},
                          std::integral_constant<int, nu>{});
                });
            });
        };
        emit(va, sa);
        emit(vb, sb);
    } else {
        static_assert(R == 16 && NB == 1, "only radix-16 passes are followed by another pass");
        const int prefix = g / Lnext;
        const int t = g % Lnext;
        auto put = [&](cf (&v)[16], cf *buf) {
            sfor<0, R>([&](auto p) {
                constexpr int pp = decltype(p)::value;
                cf val = v[rev(R, pp)];
                if constexpr (pp > 0) {
                    if constexpr (HOIST && TW::LDS_TW && S >= TW::NREG)
                        val = cmul(val, twr.ltab[((S - 1) * 15 + (pp - 1)) * TW::LSTRIDE + t % TW::LSTRIDE]);
                    else if constexpr (HOIST) val = cmul(val, twr.r[S][pp]);
                    else val = cmul(val, table[(t * pp) * PC]);
                }
                buf[padi((pp * PC + prefix) * Lnext + t) * T + col] = val;
            });
        };
        constexpr int R2 = radix_of(l, S + 1);
        constexpr int Lnn = Lnext / R2;
        constexpr int NB2 = 16 / R2;
        auto get = [&](cf (&v)[16], const cf *buf) {
            sfor<0, NB2>([&](auto u) {
                const int beta = g + NT * decltype(u)::value;
                const int pre2 = beta / Lnn;
                const int t2 = beta % Lnn;
                sfor<0, R2>([&](auto i) {
                    v[decltype(u)::value * R2 + decltype(i)::value] = buf[padi(pre2 * Lnext + t2 + Lnn * decltype(i)::value) * T + col];
                });
            });
        };
        xsync<SYNC>();                   // nobody still reads either buffer
        put(va, lds);
        bfly<R, 0>(vb);
        put(vb, lds + HALF);
        xsync<SYNC>();
        get(va, lds);
        get(vb, lds + HALF);
        fft_passes2<L, T, S + 1, HOIST, HALF, SYNC>(va, vb, lds, g, col, twr, table, sa, sb);
    }
}
