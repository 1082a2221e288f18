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
#include "field.hpp"

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

// Constraint INDEX space (fixes which power of alpha multiplies which constraint):
//   MISC 0..249, C(x) 250+128x (+2z bool, +2z+1 xor), A(j) 890+68j (+z bool, +64+l limb),
//   P(x) 2590+64x (+z), CHI(j) 2910+4j (+l), IOTA 3010..3181;  j = 5y+x.
// Evaluation SCHEDULE (which constraints one task evaluates together) is chosen for
// the GPU: constraints that read the same columns share a task, so a lane loads
// each column value once per task instead of once per constraint family:
//   task 0        MISC
//   task 1+x      S_x: C(x), A(5y+x) for y=0..4, P(x)   (c[x], c[x-1], c[x+1], c'[x], a'[.][x])
//   task 6+Y      T_Y: CHI(5Y+X) for X=0..4             (the five rho-pi planes of row Y)
//   task 11       IOTA
//   task 12       BUS: the three extension-valued LogUp constraints (indices 3182..3184)
constexpr int kNumTasks = 13;
constexpr int kBusTask = 12;
constexpr int kNumBusConstraints = 3;
constexpr int kNumAllConstraints = kNumConstraints + kNumBusConstraints;  // powers of alpha needed
constexpr int kBusTuple = 200;   // input limbs (100) || output limbs (100) of one permutation
constexpr int kPermWidth = 4;    // the running sum phi: one extension column = 4 base columns
ZKSP_HD constexpr int base_c(int x) { return 250 + 128 * x; }
ZKSP_HD constexpr int base_a(int j) { return 890 + 68 * j; }
ZKSP_HD constexpr int base_p(int x) { return 2590 + 64 * x; }
ZKSP_HD constexpr int base_chi(int j) { return 2910 + 4 * j; }
constexpr int kBaseIota = 3010;

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

// xor of boolean-valued field elements as polynomials: a + b - 2ab, and
// xor3 = xor(xor(a, b), c) = a+b+c - 2(ab+ac+bc) + 4abc (the same degree-3
// polynomial as the expanded form, two multiplications instead of four)
template <class F>
ZKSP_HD F xor2(F a, F b) {
  return a + b - (a * b).dbl();
}
template <class F>
ZKSP_HD F xor3(F a, F b, F c) {
  return xor2(xor2(a, b), c);
}

// Column of bit z of B[X,Y] = rotl(A'[(X+3Y)%5, X], rot[(X+3Y)%5][X])
ZKSP_HD int b_col(int X, int Y, int z) {
  int xa = (X + 3 * Y) % 5, ya = X;
  int rot = tables().rot[xa][ya];
  return kAp + 64 * (5 * ya + xa) + ((z + 64 - rot) & 63);
}

// Column of element j of the LogUp tuple a trace row carries: the 100 preimage limbs,
// then the 100 limbs of the round's output state (a''' for lane 0, a'' for lanes 1..24).
ZKSP_HD int bus_tuple_col(int j) {
  if (j < 100) return kPreimage + j;
  const int o = j - 100, lane = o >> 2, l = o & 3;
  return lane == 0 ? kAppp00 + l : kApp + 4 * lane + l;
}

// LogUp bus (row a6 "lookup-argument constraints"): the chip receives its tuple with
// multiplicity `export`; phi is the exclusive running sum of export / f over the rows and
// S its total, which the verifier recomputes from the public I/O list.
//   f = gamma + sum_j beta^j t_j
//   L0 = is_first * phi,  L1 = is_trans * ((phi_next - phi) f - m),  L2 = is_last * ((S - phi) f - m)
// Ctx additionally provides:  using E (extension type, E*F and E*E defined);
//   E gamma(); E beta_pow(int j); E cum_sum(); E phi_local(); E phi_next(); F is_last();
//   E lift(F); void emit_ext_at(int k, E v);
template <class Ctx>
ZKSP_HD void eval_bus(Ctx& ctx) {
  using E = typename Ctx::E;
  E f = ctx.gamma();
  for (int j = 0; j < kBusTuple; ++j) f = f + ctx.beta_pow(j) * ctx.local(bus_tuple_col(j));
  const E m = ctx.lift(ctx.local(kExport));
  const E phi = ctx.phi_local();
  ctx.emit_ext_at(kNumConstraints + 0, phi * ctx.is_first());
  ctx.emit_ext_at(kNumConstraints + 1, ((ctx.phi_next() - phi) * f - m) * ctx.is_trans());
  ctx.emit_ext_at(kNumConstraints + 2, ((ctx.cum_sum() - phi) * f - m) * ctx.is_last());
}

// Ctx interface:
//   using F;  F local(int col); F next(int col);  F is_first(); F is_trans();
//   F one();  void emit_at(int k, F v);   (adds alpha^k * v to the folded sum)
template <class Ctx>
ZKSP_HD void eval_task(int task, Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.one();
  const F zero = one - one;
  if (task == 0) {
    // ---- MISC: round flags, preimage bookkeeping, export ----
    const F is_first = ctx.is_first(), is_trans = ctx.is_trans();
    const F f0 = ctx.local(kFlags), f23 = ctx.local(kFlags + 23);
    int k = 0;
    ctx.emit_at(k++, is_first * (f0 - one));
    for (int i = 1; i < 24; ++i) ctx.emit_at(k++, is_first * ctx.local(kFlags + i));
    for (int i = 0; i < 24; ++i) ctx.emit_at(k++, is_trans * (ctx.next(kFlags + (i + 1) % 24) - ctx.local(kFlags + i)));
    for (int j = 0; j < 100; ++j) ctx.emit_at(k++, f0 * (ctx.local(kPreimage + j) - ctx.local(kA + j)));
    const F trans_nf = is_trans * (one - f23);
    for (int j = 0; j < 100; ++j) ctx.emit_at(k++, trans_nf * (ctx.next(kPreimage + j) - ctx.local(kPreimage + j)));
    const F ex = ctx.local(kExport);
    ctx.emit_at(k++, ex * (ex - one));
    ctx.emit_at(k++, (one - f23) * ex);
  } else if (task <= 5) {
    // ---- S_x: theta for column x ----
    //   C(x): c bits boolean; c' = c[x] ^ c[x-1] ^ rotl(c[x+1], 1)
    //   A(j): a' bits boolean; a limbs = recompose(a' ^ c ^ c')
    //   P(x): c'[x][z] is the parity of the a'[.][x][z] column
    const int x = task - 1;
    const int xm = (x + 4) % 5, xp = (x + 1) % 5;
    const F two = one.dbl(), four = two.dbl();
    for (int l = 3; l >= 0; --l) {
      F acc[5] = {zero, zero, zero, zero, zero};
      for (int z = 16 * l + 15; z >= 16 * l; --z) {
        const F c = ctx.local(kC + 64 * x + z);
        const F cp = ctx.local(kCp + 64 * x + z);
        ctx.emit_at(base_c(x) + 2 * z, c * (c - one));
        ctx.emit_at(base_c(x) + 2 * z + 1,
                    cp - xor3(c, ctx.local(kC + 64 * xm + z), ctx.local(kC + 64 * xp + ((z + 63) & 63))));
        const F ccp = xor2(c, cp);  // bit z of D[x] = C[x] ^ C'[x], shared by the five lanes of the column
        F s = zero;
#pragma unroll
        for (int y = 0; y < 5; ++y) {
          const int j = 5 * y + x;
          const F v = ctx.local(kAp + 64 * j + z);
          ctx.emit_at(base_a(j) + z, v * (v - one));
          acc[y] = acc[y].dbl() + xor2(v, ccp);
          s = s + v;
        }
        const F d = s - cp;
        ctx.emit_at(base_p(x) + z, d * (d - two) * (d - four));
      }
#pragma unroll
      for (int y = 0; y < 5; ++y) {
        const int j = 5 * y + x;
        ctx.emit_at(base_a(j) + 64 + l, ctx.local(kA + 4 * j + l) - acc[y]);
      }
    }
  } else if (task <= 10) {
    // ---- T_Y: chi for row Y: a'' limbs = recompose(b ^ (~b[x+1] & b[x+2])) ----
    const int Y = task - 6;
    for (int l = 0; l < 4; ++l) {
      F acc[5] = {zero, zero, zero, zero, zero};
      for (int z = 16 * l + 15; z >= 16 * l; --z) {
        F bb[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) bb[X] = ctx.local(b_col(X, Y, z));
#pragma unroll
        for (int X = 0; X < 5; ++X) {
          const F andn = (one - bb[(X + 1) % 5]) * bb[(X + 2) % 5];
          const F t = bb[X] * andn;
          acc[X] = acc[X].dbl() + (bb[X] + andn - t.dbl());
        }
      }
#pragma unroll
      for (int X = 0; X < 5; ++X) {
        const int j = 5 * Y + X;
        ctx.emit_at(base_chi(j) + l, ctx.local(kApp + 4 * j + l) - acc[X]);
      }
    }
  } else if (task == 11) {
    // ---- IOTA: a''[0][0] bits, round constant, hand-over to the next row ----
    int k = kBaseIota;
    for (int z = 0; z < 64; ++z) {
      F v = ctx.local(kApp00 + z);
      ctx.emit_at(k++, v * (v - one));
    }
    for (int l = 0; l < 4; ++l) {
      F acc = ctx.local(kApp00 + 16 * l + 15);
      for (int z = 16 * l + 14; z >= 16 * l; --z) acc = acc.dbl() + ctx.local(kApp00 + z);
      ctx.emit_at(k++, ctx.local(kApp + l) - acc);
    }
    for (int l = 0; l < 4; ++l) {
      F acc = zero;
      for (int z = 16 * l + 15; z >= 16 * l; --z) {
        F rc = zero;
        for (int r = 0; r < 24; ++r)
          if ((tables().rc[r] >> z) & 1) rc = rc + ctx.local(kFlags + r);
        F v = ctx.local(kApp00 + z);
        F t = v * rc;
        acc = acc.dbl() + (v + rc - t.dbl());
      }
      ctx.emit_at(k++, ctx.local(kAppp00 + l) - acc);
    }
    const F trans_nf = ctx.is_trans() * (one - ctx.local(kFlags + 23));
    for (int j = 0; j < 25; ++j)
      for (int l = 0; l < 4; ++l) {
        F o = (j == 0) ? ctx.local(kAppp00 + l) : ctx.local(kApp + 4 * j + l);
        ctx.emit_at(k++, trans_nf * (ctx.next(kA + 4 * j + l) - o));
      }
  }
}

}  // namespace ka
}  // namespace zksp
