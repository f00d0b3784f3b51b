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

typedef float2 cf;
#define DEVI __device__ __forceinline__

DEVI cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
DEVI cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
DEVI cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
DEVI cf cconj(cf a) { return make_float2(a.x, -a.y); }

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
    a = cadd(t, b);
    b = csub(t, b);
}
DEVI void b4(cf &a0, cf &a1, cf &a2, cf &a3) {
    cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = make_float2(t1.x - t3.y, t1.y + t3.x);  // t1 + i*t3
    a3 = make_float2(t1.x + t3.y, t1.y - t3.x);  // t1 - i*t3
}
#define MFB_H 0.70710678118654752440f
#define MFB_C 0.92387953251128675613f
#define MFB_S 0.38268343236508977173f
DEVI cf mul_w8_1(cf x) { return make_float2(MFB_H * (x.x - x.y), MFB_H * (x.x + x.y)); }   // e^{+i pi/4}
DEVI cf mul_i(cf x) { return make_float2(-x.y, x.x); }                                      // e^{+i pi/2}
DEVI cf mul_w8_3(cf x) { return make_float2(-MFB_H * (x.x + x.y), MFB_H * (x.x - x.y)); }  // e^{+i 3pi/4}
DEVI cf mul_w16_1(cf x) { return cmul(x, make_float2(MFB_C, MFB_S)); }
DEVI cf mul_w16_3(cf x) { return cmul(x, make_float2(MFB_S, MFB_C)); }
DEVI cf mul_w16_9(cf x) { return cmul(x, make_float2(-MFB_C, -MFB_S)); }

template <int R, int O>
DEVI void bfly(cf (&v)[16]) {
    if constexpr (R == 2) {
        b2(v[O], v[O + 1]);
    } else if constexpr (R == 4) {
        b4(v[O], v[O + 1], v[O + 2], v[O + 3]);
    } else if constexpr (R == 8) {
        b4(v[O + 0], v[O + 2], v[O + 4], v[O + 6]);
        b4(v[O + 1], v[O + 3], v[O + 5], v[O + 7]);
        v[O + 3] = mul_w8_1(v[O + 3]);
        v[O + 5] = mul_i(v[O + 5]);
        v[O + 7] = mul_w8_3(v[O + 7]);
        b2(v[O + 0], v[O + 1]);
        b2(v[O + 2], v[O + 3]);
        b2(v[O + 4], v[O + 5]);
        b2(v[O + 6], v[O + 7]);
    } else {
        static_assert(R == 16 && O == 0, "radix");
        b4(v[0], v[4], v[8], v[12]);
        b4(v[1], v[5], v[9], v[13]);
        b4(v[2], v[6], v[10], v[14]);
        b4(v[3], v[7], v[11], v[15]);
        // u[t][p1] sits at t + 4*p1 ; twiddle W16^(t*p1)
        v[5] = mul_w16_1(v[5]);    // t=1,p1=1
        v[9] = mul_w8_1(v[9]);     // t=1,p1=2 -> W16^2
        v[13] = mul_w16_3(v[13]);  // t=1,p1=3
        v[6] = mul_w8_1(v[6]);     // t=2,p1=1 -> W16^2
        v[10] = mul_i(v[10]);      // t=2,p1=2 -> W16^4
        v[14] = mul_w8_3(v[14]);   // t=2,p1=3 -> W16^6
        v[7] = mul_w16_3(v[7]);    // t=3,p1=1
        v[11] = mul_w8_3(v[11]);   // t=3,p1=2 -> W16^6
        v[15] = mul_w16_9(v[15]);  // t=3,p1=3 -> W16^9
        b4(v[0], v[1], v[2], v[3]);
        b4(v[4], v[5], v[6], v[7]);
        b4(v[8], v[9], v[10], v[11]);
        b4(v[12], v[13], v[14], v[15]);
    }
}

// ---- pass chain ----------------------------------------------------------------------------
// v    : the thread's 16 points; on entry of pass 0 slot i holds element g + (L/16)*i
// lds  : exchange buffer of padlen(L)*T elements; element (pos, col) at padi(pos)*T + col
// g    : thread index inside the FFT group, 0 <= g < L/16 ;  col: column inside the tile
// tw   : W_L^x = exp(+2*pi*i*x/L), x in [0, L)
// store(n, value, slot): receives natural output index n; slot is a compile-time register slot id
template <int L, int T, int S, class Store>
DEVI void fft_passes(cf (&v)[16], cf *lds, const int g, const int col, const cf *__restrict__ tw, Store &store) {
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
            const int beta = g + NT * decltype(u)::value;
            sfor<0, R>([&](auto p) {
                constexpr int slot = decltype(u)::value * R + decltype(p)::value;
                store(decltype(p)::value * PC + beta, v[decltype(u)::value * R + rev(R, decltype(p)::value)], std::integral_constant<int, slot>{});
            });
        });
    } else {
        sfor<0, NB>([&](auto u) {
            const int beta = g + NT * decltype(u)::value;
            const int prefix = beta / Lnext;
            const int t = beta % Lnext;
            sfor<0, R>([&](auto p) {
                constexpr int pp = decltype(p)::value;
                cf val = v[decltype(u)::value * R + rev(R, pp)];
                if constexpr (pp > 0) val = cmul(val, tw[(t * pp) * PC]);
                lds[padi((pp * PC + prefix) * Lnext + t) * T + col] = val;
            });
        });
        __syncthreads();
        constexpr int R2 = radix_of(l, S + 1);
        constexpr int Lnn = Lnext / R2;
        constexpr int NB2 = 16 / R2;
        sfor<0, NB2>([&](auto u) {
            const int beta = g + NT * decltype(u)::value;
            const int prefix = beta / Lnn;
            const int t = beta % Lnn;
            sfor<0, R2>([&](auto i) {
                v[decltype(u)::value * R2 + decltype(i)::value] =
                    lds[padi(prefix * Lnext + t + Lnn * decltype(i)::value) * T + col];
            });
        });
        __syncthreads();
        fft_passes<L, T, S + 1>(v, lds, g, col, tw, store);
    }
}
