// Chips of the machine proof: column layouts and base-field
// constraints, written once as field-generic templates.  The device quotient kernels instantiate
// them over Fp (one lane = one LDE-domain point), the host verifier over Fp4 (the point zeta).
//
// This is this repository's own arithmetisation of RV32IM + the keccak precompile; it stands where
// sp1-core-machine 3.4.0's CPU / memory / program / ALU / keccak-permute chips stand beneath the
// reference's `client.prove(&pk, stdin).run()` (prover/src/bin/main.rs:71-74, Cargo.lock:7130).
// What it must establish is the reference's statement: the committed guest
// (circuits/sp1-merkle-proof/src/main.rs:4-14 running crypto-ops/src/lib.rs:8-23) executed from its
// entry point to HALT with the committed public values.  DESIGN.md "Machine proof" describes the
// construction; constraint ORDER here is normative for the proof bytes.
#pragma once
#include "air_keccak.cuh"

namespace zksp {
namespace mach {

enum Chip { kCpu = 0, kKeccak, kKmem, kMemFinal, kImage, kProgram, kMul, kNumChips };

// AIR opcodes = Program-table column OP = 1 + index of the CPU selector column
enum Op {
  ADD = 1, SUB, XOR, OR, AND, SLL, SRL, SRA, SLT, SLTU, JAL, JALR, BEQ, BNE, BLT, BGE, BLTU, BGEU, LB, LH, LW, LBU, LHU,
  SB, SH, SW, MUL, MULHU, ECALL, KECCAK
};
constexpr int kNumOps = 30;
constexpr int kTsBits = 24;

// ---- CPU chip ----
constexpr int C_IS_REAL = 0, C_PC = 1, C_TS = 2, C_NEXT_PC = 3, C_OP = 4, C_WR = C_OP + kNumOps, C_USE2 = C_WR + 1,
              C_RD = C_WR + 2, C_RS1 = C_WR + 3, C_RS2 = C_WR + 4, C_IMM_LO = C_WR + 5, C_IMM_HI = C_WR + 6, C_TGT = C_WR + 7,
              C_A = C_WR + 8, C_B = C_A + 32, C_C = C_B + 32, C_M = C_C + 32, C_X = C_M + 32, C_MV_LO = C_X + 32,
              C_MV_HI = C_MV_LO + 1, C_K0 = C_MV_LO + 2, C_K1 = C_K0 + 1, C_K2 = C_K0 + 2, C_K3 = C_K0 + 3, C_EQ = C_K0 + 4,
              C_INV = C_K0 + 5, C_O0 = C_K0 + 6, C_O1 = C_O0 + 1, C_O2 = C_O0 + 2, C_O3 = C_O0 + 3, C_SC = C_O0 + 4,
              C_R1_PTS = C_SC + 6, C_R2_PTS = C_R1_PTS + 1, C_M_PTS = C_R1_PTS + 2, C_W_PTS = C_R1_PTS + 3,
              C_W_PLO = C_R1_PTS + 4, C_W_PHI = C_R1_PTS + 5, C_R1_D = C_R1_PTS + 6, C_R2_D = C_R1_D + kTsBits,
              C_M_D = C_R2_D + kTsBits, C_W_D = C_M_D + kTsBits, kCpuWidth = C_W_D + kTsBits;
enum { SC_HALT = 0, SC_WRITE, SC_COMMIT, SC_DEFER, SC_HINT_LEN, SC_HINT_READ };
static_assert(kCpuWidth == 322, "CPU chip layout");

// ---- keccak chip: p3-keccak-air's columns + the call time ----
constexpr int KC_TS = ka::kWidth, kKeccakWidth = ka::kWidth + 1;
// ---- keccak-memory chip ----
constexpr int KM_IS_REAL = 0, KM_TS = 1, KM_PTR_LO = 2, KM_PTR_HI = 3, KM_IDX = 4, KM_ISF = 5, KM_ISL = 6, KM_CALL = 7,
              KM_ADDR = 8, KM_OLD_LO = 9, KM_OLD_HI = 10, KM_NEW_LO = 11, KM_NEW_HI = 12, KM_PTS = 13, KM_D = 14,
              kKmemWidth = KM_D + kTsBits;
// ---- memory boundary chip ----
constexpr int MF_IS_REAL = 0, MF_ADDR = 1, MF_IS_INIT = 2, MF_FIN_LO = 3, MF_FIN_HI = 4, MF_FIN_TS = 5, MF_DIFF = 6,
              MF_INIT = MF_DIFF + 32, kMemFinalWidth = MF_INIT + 32;
// ---- image / program chips: preprocessed columns, one main column ----
constexpr int IMG_P_ADDR = 0, IMG_P_LO = 1, IMG_P_HI = 2, kImagePrepWidth = 3, kImageWidth = 1;
constexpr int PR_PC = 0, PR_OP = 1, PR_WR = 2, PR_USE2 = 3, PR_RD = 4, PR_RS1 = 5, PR_RS2 = 6, PR_IMM_LO = 7, PR_IMM_HI = 8,
              PR_TGT = 9, kProgramPrepWidth = 10, kProgramWidth = 1;
// ---- multiplier chip ----
constexpr int MU_IS_REAL = 0, MU_HI = 1, MU_B = 2, MU_C = MU_B + 32, MU_P = MU_C + 32, MU_Q0 = MU_P + 64, MU_Q1 = MU_Q0 + 10,
              MU_Q2 = MU_Q1 + 11, kMulWidth = MU_Q2 + 10;

enum Bus { BUS_MEM = 1, BUS_PROG, BUS_KCALL, BUS_KIO, BUS_MUL, BUS_PUBC, BUS_PUBH };

// Ctx interface:
//   using F;  F local(int col); F next(int col); F is_first(); F is_trans(); F is_last(); F pub();
//   F k(uint32_t montgomery_word)  (a constant);  void emit(F v)  (appends the next constraint)
#define ZKSP_K(c) ctx.k(cmonty(c))

template <class F, class Ctx>
ZKSP_HD F limb_of(const Ctx& ctx, int bits, int limb) {
  F s = ctx.local(bits + 16 * limb + 15);
  for (int i = 14; i >= 0; --i) s = s.dbl() + ctx.local(bits + 16 * limb + i);
  return s;
}
template <class F, class Ctx>
ZKSP_HD F byte_of(const Ctx& ctx, int bits, int byte) {
  F s = ctx.local(bits + 8 * byte + 7);
  for (int i = 6; i >= 0; --i) s = s.dbl() + ctx.local(bits + 8 * byte + i);
  return s;
}
template <class F, class Ctx>
ZKSP_HD F bits_val(const Ctx& ctx, int bits, int n) {
  F s = ctx.local(bits + n - 1);
  for (int i = n - 2; i >= 0; --i) s = s.dbl() + ctx.local(bits + i);
  return s;
}
template <class F>
ZKSP_HD F bool_c(F v, F one) {
  return v * (v - one);
}

template <class Ctx>
ZKSP_HD void eval_cpu(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1);
  const F k65536 = ZKSP_K(65536);
#define L(c) ctx.local(c)
#define OPF(op) ctx.local(C_OP + (op) - 1)
  // ---- booleans ----
  ctx.emit(bool_c(L(C_IS_REAL), one));
  for (int k = 0; k < kNumOps; ++k) ctx.emit(bool_c(L(C_OP + k), one));
  ctx.emit(bool_c(L(C_WR), one));
  ctx.emit(bool_c(L(C_USE2), one));
  for (int i = 0; i < 160; ++i) ctx.emit(bool_c(L(C_A + i), one));  // A, B, C, M, X
  for (int i = 0; i < 4; ++i) ctx.emit(bool_c(L(C_K0 + i), one));
  ctx.emit(bool_c(L(C_EQ), one));
  for (int i = 0; i < 4; ++i) ctx.emit(bool_c(L(C_O0 + i), one));
  for (int i = 0; i < 6; ++i) ctx.emit(bool_c(L(C_SC + i), one));
  for (int i = 0; i < 4 * kTsBits; ++i) ctx.emit(bool_c(L(C_R1_D + i), one));
  // ---- row structure ----
  const F is_real = L(C_IS_REAL), is_first = ctx.is_first(), is_trans = ctx.is_trans();
  {
    F opsum = L(C_OP);
    for (int k = 1; k < kNumOps; ++k) opsum = opsum + L(C_OP + k);
    ctx.emit(opsum - is_real);
    ctx.emit(L(C_WR) * (one - is_real));
    ctx.emit(L(C_USE2) * (one - is_real));
    ctx.emit(is_first * (is_real - one));
    ctx.emit(is_first * (L(C_PC) - ctx.pub()));
    ctx.emit(is_first * (L(C_TS) - ZKSP_K(4)));
    ctx.emit(is_trans * (ctx.next(C_TS) - L(C_TS) - ZKSP_K(4)));
    ctx.emit(is_trans * ctx.next(C_IS_REAL) * (ctx.next(C_PC) - L(C_NEXT_PC)));
    ctx.emit(is_trans * (ctx.next(C_IS_REAL) - is_real + L(C_SC + SC_HALT)));
    F scsum = L(C_SC);
    for (int k = 1; k < 6; ++k) scsum = scsum + L(C_SC + k);
    ctx.emit(scsum - OPF(ECALL));
  }
  // ---- limbs ----
  const F a_lo = limb_of<F>(ctx, C_A, 0), a_hi = limb_of<F>(ctx, C_A, 1), b_lo = limb_of<F>(ctx, C_B, 0),
          b_hi = limb_of<F>(ctx, C_B, 1), c_lo = limb_of<F>(ctx, C_C, 0), c_hi = limb_of<F>(ctx, C_C, 1),
          m_lo = limb_of<F>(ctx, C_M, 0), m_hi = limb_of<F>(ctx, C_M, 1), x_lo = limb_of<F>(ctx, C_X, 0),
          x_hi = limb_of<F>(ctx, C_X, 1);
  const F k0 = L(C_K0), k1 = L(C_K1), k2 = L(C_K2), k3 = L(C_K3);
  // ---- operand C is the immediate ----
  {
    const F immc = is_real - L(C_USE2);
    ctx.emit(immc * (c_lo - L(C_IMM_LO)));
    ctx.emit(immc * (c_hi - L(C_IMM_HI)));
  }
  // ---- add / sub ----
  ctx.emit(OPF(ADD) * (b_lo + c_lo - (a_lo + k65536 * k0)));
  ctx.emit(OPF(ADD) * (b_hi + c_hi + k0 - (a_hi + k65536 * k1)));
  ctx.emit(OPF(SUB) * (a_lo + c_lo - (b_lo + k65536 * k0)));
  ctx.emit(OPF(SUB) * (a_hi + c_hi + k0 - (b_hi + k65536 * k1)));
  // ---- bitwise ----
  for (int op = XOR; op <= AND; ++op)
    for (int h = 0; h < 2; ++h) {
      F acc = one - one;
      for (int i = 15; i >= 0; --i) {
        const F b = L(C_B + 16 * h + i), c = L(C_C + 16 * h + i), bc = b * c;
        const F bit = op == AND ? bc : op == OR ? b + c - bc : b + c - bc.dbl();
        acc = acc.dbl() + bit;
      }
      ctx.emit(OPF(op) * ((h ? a_hi : a_lo) - acc));
    }
  // ---- shifts: X is the one-hot of the amount ----
  {
    const F sh = OPF(SLL) + OPF(SRL) + OPF(SRA);
    F sum = L(C_X), idx = one - one;
    for (int k = 1; k < 32; ++k) {
      sum = sum + L(C_X + k);
      idx = idx + ctx.k(cmonty((uint32_t)k)) * L(C_X + k);
    }
    ctx.emit(sh * (sum - one));
    ctx.emit(sh * (idx - bits_val<F>(ctx, C_C, 5)));
    for (int kind = 0; kind < 3; ++kind) {
      const F sel = OPF(kind == 0 ? SLL : kind == 1 ? SRL : SRA);
      for (int h = 0; h < 2; ++h) {
        F acc = one - one;
        for (int i = 15; i >= 0; --i) {
          const int j = 16 * h + i;
          F t = one - one;
          for (int k = 0; k < 32; ++k) {
            int src;
            if (kind == 0) { if (k > j) continue; src = j - k; }
            else if (kind == 1) { if (j + k > 31) continue; src = j + k; }
            else src = j + k > 31 ? 31 : j + k;
            t = t + L(C_X + k) * L(C_B + src);
          }
          acc = acc.dbl() + t;
        }
        ctx.emit(sel * ((h ? a_hi : a_lo) - acc));
      }
    }
  }
  // ---- comparisons: X = B - C (sign bits flipped for the signed orders), K1 = "less than" ----
  {
    const F sgn = OPF(SLT) + OPF(BLT) + OPF(BGE);
    const F cmp = OPF(SLT) + OPF(SLTU) + OPF(BEQ) + OPF(BNE) + OPF(BLT) + OPF(BGE) + OPF(BLTU) + OPF(BGEU);
    ctx.emit(cmp * (b_lo - c_lo + k65536 * k0 - x_lo));
    ctx.emit(cmp * (b_hi - c_hi - k0 + k65536 * k1 - x_hi) + k65536 * (sgn * (L(C_C + 31) - L(C_B + 31))));
    const F bq = OPF(BEQ) + OPF(BNE), z = x_lo + x_hi;
    ctx.emit(bq * (z * L(C_INV) - one + L(C_EQ)));
    ctx.emit(bq * (z * L(C_EQ)));
    const F slt = OPF(SLT) + OPF(SLTU);
    ctx.emit(slt * (a_lo - k1));
    ctx.emit(slt * a_hi);
  }
  // ---- next pc ----
  {
    const F pc4 = L(C_PC) + ZKSP_K(4), np = L(C_NEXT_PC), tgt = L(C_TGT);
    const F def = is_real - OPF(JAL) - OPF(JALR) - OPF(BEQ) - OPF(BNE) - OPF(BLT) - OPF(BGE) - OPF(BLTU) - OPF(BGEU) - OPF(KECCAK);
    ctx.emit(def * (np - pc4));
    ctx.emit(OPF(JAL) * (np - tgt));
    ctx.emit(OPF(JAL) * (a_lo - c_lo));
    ctx.emit(OPF(JAL) * (a_hi - c_hi));
    ctx.emit(OPF(JALR) * (a_lo + k65536 * a_hi - tgt));
    ctx.emit(OPF(JALR) * (np - (x_lo + k65536 * x_hi - L(C_X))));
    const F eq = L(C_EQ), d = tgt - pc4, base = np - pc4;
    ctx.emit(OPF(BEQ) * (base - eq * d));
    ctx.emit(OPF(BNE) * (base - (one - eq) * d));
    ctx.emit(OPF(BLT) * (base - k1 * d));
    ctx.emit(OPF(BGE) * (base - (one - k1) * d));
    ctx.emit(OPF(BLTU) * (base - k1 * d));
    ctx.emit(OPF(BGEU) * (base - (one - k1) * d));
    ctx.emit(OPF(KECCAK) * (np - (b_lo + k65536 * b_hi)));
  }
  // ---- address adder: X = B + imm ----
  const F loads = OPF(LB) + OPF(LH) + OPF(LW) + OPF(LBU) + OPF(LHU), stores = OPF(SB) + OPF(SH) + OPF(SW);
  {
    const F ad = loads + stores + OPF(JALR);
    ctx.emit(ad * (b_lo + L(C_IMM_LO) - (x_lo + k65536 * k2)));
    ctx.emit(ad * (b_hi + L(C_IMM_HI) + k2 - (x_hi + k65536 * k3)));
  }
  // ---- byte offset one-hot ----
  const F o0 = L(C_O0), o1 = L(C_O1), o2 = L(C_O2), o3 = L(C_O3);
  {
    const F ls = loads + stores;
    ctx.emit(ls * (o0 + o1 + (o2 + o3) - one));
    ctx.emit(ls * (o1 + o2.dbl() + ZKSP_K(3) * o3 - (L(C_X) + L(C_X + 1).dbl())));
    ctx.emit(OPF(ECALL) * (o0 - one));
    ctx.emit(OPF(ECALL) * (o1 + o2 + o3));
    ctx.emit(OPF(ECALL) * (x_lo - ZKSP_K(11)));
    ctx.emit(OPF(ECALL) * x_hi);
  }
  // ---- loads and stores ----
  {
    const F mb[4] = {byte_of<F>(ctx, C_M, 0), byte_of<F>(ctx, C_M, 1), byte_of<F>(ctx, C_M, 2), byte_of<F>(ctx, C_M, 3)};
    const F k65535 = ZKSP_K(65535);
    ctx.emit(OPF(LW) * (o0 - one));
    ctx.emit(OPF(LW) * (a_lo - m_lo));
    ctx.emit(OPF(LW) * (a_hi - m_hi));
    const F hv = o0 * m_lo + o2 * m_hi, hs = o0 * L(C_M + 15) + o2 * L(C_M + 31);
    ctx.emit(OPF(LHU) * (o1 + o3));
    ctx.emit(OPF(LHU) * (a_lo - hv));
    ctx.emit(OPF(LHU) * a_hi);
    ctx.emit(OPF(LH) * (o1 + o3));
    ctx.emit(OPF(LH) * (a_lo - hv));
    ctx.emit(OPF(LH) * (a_hi - k65535 * hs));
    F bv = one - one, bs = one - one;
    for (int p = 0; p < 4; ++p) {
      bv = bv + L(C_O0 + p) * mb[p];
      bs = bs + L(C_O0 + p) * L(C_M + 8 * p + 7);
    }
    ctx.emit(OPF(LBU) * (a_lo - bv));
    ctx.emit(OPF(LBU) * a_hi);
    ctx.emit(OPF(LB) * (a_lo - (bv + ZKSP_K(0xff00) * bs)));
    ctx.emit(OPF(LB) * (a_hi - k65535 * bs));
    const F keep = loads + OPF(ECALL);
    ctx.emit(keep * (L(C_MV_LO) - m_lo));
    ctx.emit(keep * (L(C_MV_HI) - m_hi));
    ctx.emit(OPF(SW) * (o0 - one));
    ctx.emit(OPF(SW) * (L(C_MV_LO) - c_lo));
    ctx.emit(OPF(SW) * (L(C_MV_HI) - c_hi));
    ctx.emit(OPF(SH) * (o1 + o3));
    ctx.emit(OPF(SH) * (L(C_MV_LO) - m_lo - o0 * (c_lo - m_lo)));
    ctx.emit(OPF(SH) * (L(C_MV_HI) - m_hi - o2 * (c_lo - m_hi)));
    const F cb = byte_of<F>(ctx, C_C, 0), k256 = ZKSP_K(256);
    ctx.emit(OPF(SB) * (L(C_MV_LO) - m_lo - (o0 * (cb - mb[0]) + k256 * (o1 * (cb - mb[1])))));
    ctx.emit(OPF(SB) * (L(C_MV_HI) - m_hi - (o2 * (cb - mb[2]) + k256 * (o3 * (cb - mb[3])))));
  }
  // ---- ecall ----
  {
    const F code = ZKSP_K(0x02) * L(C_SC + SC_WRITE) + ZKSP_K(0x10) * L(C_SC + SC_COMMIT) + ZKSP_K(0x1a) * L(C_SC + SC_DEFER) +
                   ZKSP_K(0xf0) * L(C_SC + SC_HINT_LEN) + ZKSP_K(0xf1) * L(C_SC + SC_HINT_READ);
    ctx.emit(OPF(ECALL) * (b_lo - code));
    ctx.emit(OPF(ECALL) * b_hi);
    const F same = OPF(ECALL) - L(C_SC + SC_HINT_LEN);
    ctx.emit(same * (a_lo - b_lo));
    ctx.emit(same * (a_hi - b_hi));
  }
  // ---- previous access times are older ----
  {
    const F memq = loads + stores + OPF(ECALL), ts = L(C_TS);
    ctx.emit(is_real * (ts - L(C_R1_PTS) - one - bits_val<F>(ctx, C_R1_D, kTsBits)));
    ctx.emit(L(C_USE2) * (ts - L(C_R2_PTS) - bits_val<F>(ctx, C_R2_D, kTsBits)));
    ctx.emit(memq * (ts + one - L(C_M_PTS) - bits_val<F>(ctx, C_M_D, kTsBits)));
    ctx.emit(L(C_WR) * (ts + ZKSP_K(2) - L(C_W_PTS) - bits_val<F>(ctx, C_W_D, kTsBits)));
  }
#undef OPF
}
constexpr int kCpuConstraints = 392;

template <class Ctx>
ZKSP_HD void eval_kmem(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), is_first = ctx.is_first(), is_trans = ctx.is_trans();
  ctx.emit(bool_c(L(KM_IS_REAL), one));
  ctx.emit(bool_c(L(KM_ISF), one));
  ctx.emit(bool_c(L(KM_ISL), one));
  for (int i = 0; i < kTsBits; ++i) ctx.emit(bool_c(L(KM_D + i), one));
  ctx.emit(L(KM_CALL) - L(KM_ISF) * L(KM_IS_REAL));
  ctx.emit(is_first * L(KM_IDX));
  ctx.emit(is_first * (L(KM_ISF) - one));
  const F nl = one - L(KM_ISL);
  ctx.emit(is_trans * (ctx.next(KM_IDX) - (L(KM_IDX) + one) * nl));
  ctx.emit(L(KM_ISL) * (L(KM_IDX) - ZKSP_K(49)));
  ctx.emit(is_trans * (ctx.next(KM_ISF) - L(KM_ISL)));
  ctx.emit(is_trans * nl * (ctx.next(KM_IS_REAL) - L(KM_IS_REAL)));
  ctx.emit(is_trans * ctx.next(KM_IS_REAL) * (one - L(KM_IS_REAL)));
  ctx.emit(is_trans * nl * (ctx.next(KM_TS) - L(KM_TS)));
  ctx.emit(is_trans * nl * (ctx.next(KM_PTR_LO) - L(KM_PTR_LO)));
  ctx.emit(is_trans * nl * (ctx.next(KM_PTR_HI) - L(KM_PTR_HI)));
  ctx.emit(L(KM_IS_REAL) * (L(KM_ADDR) - (L(KM_PTR_LO) + ZKSP_K(65536) * L(KM_PTR_HI) + ZKSP_K(4) * L(KM_IDX))));
  ctx.emit(L(KM_IS_REAL) * (L(KM_TS) + one - L(KM_PTS) - bits_val<F>(ctx, KM_D, kTsBits)));
}
constexpr int kKmemConstraints = 40;

template <class Ctx>
ZKSP_HD void eval_memfinal(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1);
  ctx.emit(bool_c(L(MF_IS_REAL), one));
  ctx.emit(bool_c(L(MF_IS_INIT), one));
  for (int i = 0; i < 64; ++i) ctx.emit(bool_c(L(MF_DIFF + i), one));  // DIFF, INIT
  ctx.emit(L(MF_IS_INIT) * (one - L(MF_IS_REAL)));
  const F tn = ctx.is_trans() * ctx.next(MF_IS_REAL);
  ctx.emit(tn * (one - L(MF_IS_REAL)));
  ctx.emit(tn * (ctx.next(MF_ADDR) - L(MF_ADDR) - one - bits_val<F>(ctx, MF_DIFF, 32)));
}
constexpr int kMemFinalConstraints = 69;

template <class Ctx>
ZKSP_HD void eval_mul(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1);
  ctx.emit(bool_c(L(MU_IS_REAL), one));
  ctx.emit(bool_c(L(MU_HI), one));
  for (int i = 0; i < 32 + 32 + 64 + 31; ++i) ctx.emit(bool_c(L(MU_B + i), one));
  ctx.emit(L(MU_HI) * (one - L(MU_IS_REAL)));
  F b[4], c[4], sk[7];
  for (int i = 0; i < 4; ++i) {
    b[i] = byte_of<F>(ctx, MU_B, i);
    c[i] = byte_of<F>(ctx, MU_C, i);
  }
  for (int k = 0; k < 7; ++k) sk[k] = one - one;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) sk[i + j] = sk[i + j] + b[i] * c[j];
  const F q0 = bits_val<F>(ctx, MU_Q0, 10), q1 = bits_val<F>(ctx, MU_Q1, 11), q2 = bits_val<F>(ctx, MU_Q2, 10);
  const F k256 = ZKSP_K(256), k65536 = ZKSP_K(65536);
  ctx.emit(sk[0] + k256 * sk[1] - (limb_of<F>(ctx, MU_P, 0) + k65536 * q0));
  ctx.emit(sk[2] + k256 * sk[3] + q0 - (limb_of<F>(ctx, MU_P, 1) + k65536 * q1));
  ctx.emit(sk[4] + k256 * sk[5] + q1 - (limb_of<F>(ctx, MU_P, 2) + k65536 * q2));
  ctx.emit(sk[6] + q2 - limb_of<F>(ctx, MU_P, 3));
}
constexpr int kMulConstraints = 166;

template <class Ctx>
ZKSP_HD void eval_image(Ctx& ctx) {
  ctx.emit(bool_c(L(0), ctx.k(kR1)));
}
// the keccak chip's extra constraint after p3-keccak-air's 3182: the call time is constant inside
// a permutation's 24 rows
template <class Ctx>
ZKSP_HD void eval_keccak_ts(Ctx& ctx) {
  ctx.emit(ctx.is_trans() * (ctx.k(kR1) - L(ka::kFlags + 23)) * (ctx.next(KC_TS) - L(KC_TS)));
}
constexpr int kKeccakConstraints = ka::kNumConstraints + 1;
#undef L

ZKSP_HD constexpr int num_constraints(int chip) {
  return chip == kCpu ? kCpuConstraints : chip == kKeccak ? kKeccakConstraints : chip == kKmem ? kKmemConstraints
       : chip == kMemFinal ? kMemFinalConstraints : chip == kImage ? 1 : chip == kProgram ? 0 : kMulConstraints;
}

}  // namespace mach
}  // namespace zksp
