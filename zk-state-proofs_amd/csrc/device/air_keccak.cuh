// keccak-f[1600] precompile-chip AIR: column layout and the constraint set,
// written once as a field-generic template.  The device quotient kernel
// instantiates it over Fp (one lane = one LDE-domain point) and the host verifier
// over Fp4 (the out-of-domain point zeta), so prover and verifier cannot drift.
//
// Replaces p3-keccak-air 0.1.4-succinct (reference Cargo.lock:5283) as wrapped by
// sp1-core-machine's keccak-permute chip (Cargo.lock:7130): 24 rows per
// permutation, 2633 columns, 64-bit lanes as 4 x u16 limbs, degree-3 constraints.
// Constraint ORDER and grouping are this repository's own (DESIGN.md "Keccak AIR").
#pragma once
#include "field.cuh"

namespace zksp {
namespace ka {

constexpr int kFlags = 0;      // step_flags[24]
constexpr int kExport = 24;    // export
constexpr int kPreimage = 25;  // preimage[y][x][limb]   (lane j = 5y+x)
constexpr int kA = 125;        // a[y][x][limb]
constexpr int kC = 225;        // c[x][z]
constexpr int kCp = 545;       // c'[x][z]
constexpr int kAp = 865;       // a'[y][x][z]
constexpr int kApp = 2465;     // a''[y][x][limb]
constexpr int kApp00 = 2565;   // a''[0][0] bits
constexpr int kAppp00 = 2629;  // a'''[0][0] limbs
constexpr int kWidth = 2633;
constexpr int kNumConstraints = 3182;
// Columns of the NEXT row that any constraint reads: flags, preimage, a.
constexpr int kNextCols = 225;

// Constraint groups (one GPU task each): MISC, C(x) x5, A(j) x25, P(x) x5,
// CHI(j) x25, IOTA.
constexpr int kNumGroups = 62;
ZKSP_HD constexpr int group_base(int g) {
  return g == 0 ? 0 : g <= 5 ? 250 + 128 * (g - 1) : g <= 30 ? 890 + 68 * (g - 6) : g <= 35 ? 2590 + 64 * (g - 31)
                                                              : g <= 60 ? 2910 + 4 * (g - 36) : 3010;
}
ZKSP_HD constexpr int group_size(int g) {
  return g == 0 ? 250 : g <= 5 ? 128 : g <= 30 ? 68 : g <= 35 ? 64 : g <= 60 ? 4 : 172;
}

struct Tables {
  uint64_t rc[24];
  uint8_t rot[5][5];  // rotation offset of lane (x, y), indexed [x][y]
};
__host__ __device__ inline const Tables& tables() {
  static constexpr Tables t = {
      {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
       0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
       0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
       0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
       0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
       0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull},
      {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}}};
  return t;
}

// xor of three boolean-valued field elements as a degree-3 polynomial
template <class F>
ZKSP_HD F xor3(F a, F b, F c) {
  F ab = a * b;
  F s2 = ab + a * c + b * c;
  F abc = ab * c;
  return a + b + c - s2.dbl() + abc.dbl().dbl();
}

// Column of bit z of B[X,Y] = rotl(A'[(X+3Y)%5, X], rot[(X+3Y)%5][X])
ZKSP_HD int b_col(int X, int Y, int z) {
  int xa = (X + 3 * Y) % 5, ya = X;
  int rot = tables().rot[xa][ya];
  return kAp + 64 * (5 * ya + xa) + ((z + 64 - rot) & 63);
}

// Ctx interface:
//   using F;  F local(int col); F next(int col);  F is_first(); F is_trans();
//   F one();  void emit(F v);   (emit folds v with the next power of alpha)
template <class Ctx>
ZKSP_HD void eval_group(int g, Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.one();
  if (g == 0) {
    // ---- MISC: round flags, preimage bookkeeping, export ----
    const F is_first = ctx.is_first(), is_trans = ctx.is_trans();
    const F f0 = ctx.local(kFlags), f23 = ctx.local(kFlags + 23);
    ctx.emit(is_first * (f0 - one));
    for (int i = 1; i < 24; ++i) ctx.emit(is_first * ctx.local(kFlags + i));
    for (int i = 0; i < 24; ++i) ctx.emit(is_trans * (ctx.next(kFlags + (i + 1) % 24) - ctx.local(kFlags + i)));
    for (int j = 0; j < 100; ++j) ctx.emit(f0 * (ctx.local(kPreimage + j) - ctx.local(kA + j)));
    const F trans_nf = is_trans * (one - f23);
    for (int j = 0; j < 100; ++j) ctx.emit(trans_nf * (ctx.next(kPreimage + j) - ctx.local(kPreimage + j)));
    const F ex = ctx.local(kExport);
    ctx.emit(ex * (ex - one));
    ctx.emit((one - f23) * ex);
  } else if (g <= 5) {
    // ---- C(x): c bits boolean; c' = c[x] ^ c[x-1] ^ rotl(c[x+1], 1) ----
    const int x = g - 1;
    const int xm = (x + 4) % 5, xp = (x + 1) % 5;
    for (int z = 0; z < 64; ++z) {
      F c = ctx.local(kC + 64 * x + z);
      ctx.emit(c * (c - one));
      F t = xor3(c, ctx.local(kC + 64 * xm + z), ctx.local(kC + 64 * xp + ((z + 63) & 63)));
      ctx.emit(ctx.local(kCp + 64 * x + z) - t);
    }
  } else if (g <= 30) {
    // ---- A(j): a' bits boolean; a limbs = recompose(a' ^ c ^ c') ----
    const int j = g - 6, x = j % 5;
    for (int z = 0; z < 64; ++z) {
      F v = ctx.local(kAp + 64 * j + z);
      ctx.emit(v * (v - one));
    }
    for (int l = 0; l < 4; ++l) {
      F acc = xor3(ctx.local(kAp + 64 * j + 16 * l + 15), ctx.local(kC + 64 * x + 16 * l + 15),
                   ctx.local(kCp + 64 * x + 16 * l + 15));
      for (int z = 16 * l + 14; z >= 16 * l; --z)
        acc = acc.dbl() + xor3(ctx.local(kAp + 64 * j + z), ctx.local(kC + 64 * x + z), ctx.local(kCp + 64 * x + z));
      ctx.emit(ctx.local(kA + 4 * j + l) - acc);
    }
  } else if (g <= 35) {
    // ---- P(x): c'[x][z] is the parity of the a'[.][x][z] column ----
    const int x = g - 31;
    const F two = one.dbl(), four = two.dbl();
    for (int z = 0; z < 64; ++z) {
      F s = ctx.local(kAp + 64 * x + z);
      for (int y = 1; y < 5; ++y) s = s + ctx.local(kAp + 64 * (5 * y + x) + z);
      F d = s - ctx.local(kCp + 64 * x + z);
      ctx.emit(d * (d - two) * (d - four));
    }
  } else if (g <= 60) {
    // ---- CHI(j): a'' limbs = recompose(b ^ (~b[x+1] & b[x+2])) ----
    const int j = g - 36, X = j % 5, Y = j / 5;
    for (int l = 0; l < 4; ++l) {
      F acc = one - one;
      for (int z = 16 * l + 15; z >= 16 * l; --z) {
        F b0 = ctx.local(b_col(X, Y, z));
        F b1 = ctx.local(b_col((X + 1) % 5, Y, z));
        F b2 = ctx.local(b_col((X + 2) % 5, Y, z));
        F andn = (one - b1) * b2;
        F t = b0 * andn;
        acc = acc.dbl() + (b0 + andn - t.dbl());
      }
      ctx.emit(ctx.local(kApp + 4 * j + l) - acc);
    }
  } else {
    // ---- IOTA: a''[0][0] bits, round constant, hand-over to the next row ----
    for (int z = 0; z < 64; ++z) {
      F v = ctx.local(kApp00 + z);
      ctx.emit(v * (v - one));
    }
    for (int l = 0; l < 4; ++l) {
      F acc = ctx.local(kApp00 + 16 * l + 15);
      for (int z = 16 * l + 14; z >= 16 * l; --z) acc = acc.dbl() + ctx.local(kApp00 + z);
      ctx.emit(ctx.local(kApp + l) - acc);
    }
    for (int l = 0; l < 4; ++l) {
      F acc = one - one;
      for (int z = 16 * l + 15; z >= 16 * l; --z) {
        F rc = one - one;
        for (int r = 0; r < 24; ++r)
          if ((tables().rc[r] >> z) & 1) rc = rc + ctx.local(kFlags + r);
        F v = ctx.local(kApp00 + z);
        F t = v * rc;
        acc = acc.dbl() + (v + rc - t.dbl());
      }
      ctx.emit(ctx.local(kAppp00 + l) - acc);
    }
    const F trans_nf = ctx.is_trans() * (one - ctx.local(kFlags + 23));
    for (int j = 0; j < 25; ++j)
      for (int l = 0; l < 4; ++l) {
        F o = (j == 0) ? ctx.local(kAppp00 + l) : ctx.local(kApp + 4 * j + l);
        ctx.emit(trans_nf * (ctx.next(kA + 4 * j + l) - o));
      }
  }
}

}  // namespace ka
}  // namespace zksp
