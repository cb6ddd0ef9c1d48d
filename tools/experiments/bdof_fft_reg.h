// 512-point line transforms of ONE wave with the stage exchanges inside the wave's registers (gfx950).
//
// line_fft (bdof_fft.h) is a Stockham autosort transform: after every radix-8 stage the wave writes its 8 points per lane to
// its LDS row image and reads them back in the next stage's order — two LDS round trips per transform.  Here the same three
// radix-8 stages run as decimation in frequency with the data exchanged between REGISTER INDEX and LANE INDEX bits instead:
//
//   position n = 64 n2 + 8 n1 + n0  (n2 = register, lane = 8 n1 + n0)        frequency k = k0 + 8 k1 + 64 k2
//   A : DFT8 over the register      (n2 -> k0),  twiddle W512^(lane k0)
//   E1: register bits <-> lane bits 5:3           register = n1, lane = (k0, n0)
//   B : DFT8 over the register      (n1 -> k1),  twiddle W64^(n0 k1)
//   E2: register bits <-> lane bits 2:0           register = n0, lane = (k0, k1)
//   C : DFT8 over the register      (n0 -> k2)                               result X[k] in register k2 of lane 8 k0 + k1
//
// The spectrum comes out in that digit-permuted order, which a point-wise multiplier does not mind (its table is laid out to
// match: fft512_perm), and the inverse transform runs the mirror image C* E2 B* E1 A* from the permuted order back to natural
// order — the classic DIF / DIT pairing of a fast convolution, so no reordering pass exists at all.
//
// E1 / E2 are 8 x 8 transposes between 3 register-index bits and 3 lane-index bits, three rounds of 2 x 2 block swaps each:
//   lane bit 5: v_permlane32_swap   (one instruction per register pair and dword)
//   lane bit 4: v_permlane16_swap   (one)
//   lane bit 3: v_mov_dpp row_ror:8 with bank masks (two)
//   lane bit 2: v_mov_dpp row_shr:4 / row_shl:4 with bank masks (two)
//   lane bits 1, 0: v_mov_dpp quad_perm + v_cndmask (four)
#pragma once
#include "../../beyond_dof_amd/csrc/bdof_fft.h"

__device__ __forceinline__ void xl_swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xl_swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// a: register whose index bit is 0, b: whose bit is 1.  Lanes whose lane bit is 1 take their new a from the partner's b,
// lanes whose bit is 0 their new b from the partner's a (partner = lane with that bit flipped).
template <int CTRL_A, int BANK_A, int CTRL_B, int BANK_B> __device__ __forceinline__ void xl_dpp(float& a, float& b) {
    const unsigned ta = __float_as_uint(a), tb = __float_as_uint(b);
    a = __uint_as_float(__builtin_amdgcn_update_dpp(ta, tb, CTRL_A, 0xf, BANK_A, false));
    b = __uint_as_float(__builtin_amdgcn_update_dpp(tb, ta, CTRL_B, 0xf, BANK_B, false));
}
template <int QP> __device__ __forceinline__ void xl_quad(float& a, float& b, bool hi) {
    const float pb = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(b), QP, 0xf, 0xf, true));
    const float pa = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), QP, 0xf, 0xf, true));
    a = hi ? pb : a;
    b = hi ? b : pa;
}

template <int LANEBIT> __device__ __forceinline__ void xl_pair(cf& a, cf& b, int lane) {
    if constexpr (LANEBIT == 5) { xl_swap32(a.x, b.x); xl_swap32(a.y, b.y); }
    else if constexpr (LANEBIT == 4) { xl_swap16(a.x, b.x); xl_swap16(a.y, b.y); }
    else if constexpr (LANEBIT == 3) { xl_dpp<0x128, 0xC, 0x128, 0x3>(a.x, b.x); xl_dpp<0x128, 0xC, 0x128, 0x3>(a.y, b.y); }      // row_ror:8
    else if constexpr (LANEBIT == 2) { xl_dpp<0x114, 0xA, 0x104, 0x5>(a.x, b.x); xl_dpp<0x114, 0xA, 0x104, 0x5>(a.y, b.y); }      // row_shr:4 / row_shl:4
    else if constexpr (LANEBIT == 1) { const bool hi = lane & 2; xl_quad<0x4E>(a.x, b.x, hi); xl_quad<0x4E>(a.y, b.y, hi); }      // quad_perm:[2,3,0,1]
    else { const bool hi = lane & 1; xl_quad<0xB1>(a.x, b.x, hi); xl_quad<0xB1>(a.y, b.y, hi); }                                   // quad_perm:[1,0,3,2]
}

// register index bits (2, 1, 0) <-> lane bits (HI + 2, HI + 1, HI)
template <int HI> __device__ __forceinline__ void xl_transpose8(cf (&u)[8], int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) xl_pair<HI + 2>(u[r], u[r + 4], lane);
#pragma unroll
    for (int r = 0; r < 8; r += 4) { xl_pair<HI + 1>(u[r], u[r + 2], lane); xl_pair<HI + 1>(u[r + 1], u[r + 3], lane); }
#pragma unroll
    for (int r = 0; r < 8; r += 2) xl_pair<HI>(u[r], u[r + 1], lane);
}

// index of the frequency that register m of lane `lane` holds after fft512_reg_forward
__host__ __device__ __forceinline__ int fft512_perm(int lane, int m) { return (lane >> 3) + 8 * (lane & 7) + 64 * m; }

// natural order in (u[m] <-> position lane + 64 m), digit-permuted spectrum out (fft512_perm).  tw: the row kernels' tables
// (w[m-1] = W512^(m lane) in registers; the middle stage's LDS table [k][m-1] = W64^(m k)).  Un-normalised, forward sign.
template <int ROUND = 1, bool EX = BDOF_EX_ALL>
__device__ __forceinline__ void fft512_reg_forward(cf (&u)[8], const FftTw<512>& tw, int lane) {
    typedef FftTw<512> TW;
    dft8<-1, EX ? 0 : ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], tw.sq);
#pragma unroll
    for (int m = 1; m < 8; ++m) u[m] = tw_mul<-1, EX>(u[m], tw.w[m - 1], EX ? tw.tail[(7 + m - 1) * 64 + lane] : tw.w[m - 1]);
    xl_transpose8<3>(u, lane);
    dft8<-1, EX ? 0 : ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], tw.sq);
    const cf* row = tw.mid + TW::M1_OFF + (lane & 7) * 7;
#pragma unroll
    for (int m = 1; m < 8; ++m) u[m] = tw_mul<-1, EX>(u[m], row[m - 1], row[(EX ? TW::LDS_CNT : 0) + m - 1]);
    xl_transpose8<0>(u, lane);
    dft8<-1, EX ? 0 : ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], tw.sq);
}

// digit-permuted order in, the first two inverse stages in registers; the LAST stage's inputs are left in the line's LDS image
// exactly where line_fft_partial<512, +1> leaves them (position j + 64 m for input m of butterfly j), for transposed_tail.
template <int ROUND = 1, bool EX = BDOF_EX_ALL, class L>
__device__ __forceinline__ void fft512_reg_inverse_partial(cf (&u)[8], const FftTw<512>& tw, int lane, L& lds) {
    typedef FftTw<512> TW;
    dft8<+1, EX ? 0 : ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], tw.sq);
    xl_transpose8<0>(u, lane);                                 // register = k1, lane = (k0, n0)
    const cf* row = tw.mid + TW::M1_OFF + (lane & 7) * 7;
#pragma unroll
    for (int m = 1; m < 8; ++m) u[m] = tw_mul<+1, EX>(u[m], row[m - 1], row[(EX ? TW::LDS_CNT : 0) + m - 1]);
    dft8<+1, EX ? 0 : ROUND>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], tw.sq);       // register = n1
    // position 64 k0 + 8 n1 + n0: the exchange E1 happens in the write
    const int bs = lds.slot(64 * (lane >> 3) + (lane & 7));
#pragma unroll
    for (int m = 0; m < 8; ++m) lds.st_at(bs, 8 * m, u[m]);
}
