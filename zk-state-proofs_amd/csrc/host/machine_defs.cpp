// See machine_defs.hpp.  Bus protocol (DESIGN.md "Machine proof", format v12):
//   MEM    (addr, lo, hi, time)   every access consumes its predecessor's tuple and produces its own
//   PROG   (pc, class, code, uc, wr, use2, rd, rs1, rs2, imm_lo, imm_hi, tgt_lo, tgt_hi)   instruction fetch, every CPU row
//   KCALL  (time, ptr_lo, ptr_hi)   CPU -> keccak-memory: one precompile call
//   KIO    (time, word index, in_lo, in_hi, out_lo, out_hi)   keccak-f chip -> keccak-memory
//   ALU    (op, a_lo, a_hi, b_lo, b_hi, c_lo, c_hi)   CPU -> ALU chip / bitwise chip / multiplier
//   SUB    (op, byte offset, a_lo, a_hi, m_lo, m_hi, c_lo, mv_lo, mv_hi)   CPU -> sub-word chip
//   PUBC   (kind, index, lo, hi), PUBH (exit_lo, exit_hi)   CPU -> verifier
//   RANGE  (kind, value)   value < 2^16 (kind 0), and a multiple of 4 (kind 1), or in 1 .. 0x77FE (kind 2): table chip
//   BYTES  (x, y)   two bytes: table chip
//   BYTEOP (kind, x, y, z)   z = x xor y (1), x or y (2), x and y (3): table chip <- bitwise chip
//   IMG    (addr, lo, hi)   image chip -> memory boundary: the initial value of an image address
//   DIGEST (tag, type, key, mask, 8 words)   Poseidon2 chip: heap nodes (type 2: children in, parent out; verifier: the leaves
//          in, the root out), injected row hashes (type 1), the ends of the openings of a leaf proof (type 0: -> verifier)
//   PAIR   (tag, lo[4], hi[4])   Poseidon2 chip (a FRI leaf's sponge row) -> query chip
//   POS    (tag, key, mask, root id)   Poseidon2 chip (the end of a run) -> query chip: where the opening sits
//   ROOT   (root id, 8 words)   transcript chip (the row that absorbed the root; the verifier for the preprocessed root) ->
//          Poseidon2 chip (the end of a run)
//   SEG    (tag, key, mask, sum[4], alpha_f[4])   Poseidon2 chip (the last block of a matrix row's hash) -> query chip
//   TBLK   (leaf, step, flags, root id / layer + 4, 8 words), TSQ (leaf, step, flags, first query)   verifier -> transcript chip
//   FINAL  (leaf, value[4]), ZETA (leaf, zeta[4]), AF (leaf, alpha_f[4], delta[4]), BETA (leaf, layer, beta[4])
//          transcript chip -> query chip
//   POW    (leaf, word)   transcript chip -> verifier: the proof-of-work sample
//   QIDX   (leaf, query, word)   transcript chip -> query chip: the word a query's index is the low bits of
//   LEAFK  (leaf, last layer, 1 / omega), BCONST (leaf, layer, has preprocessed, w_H, B1[4], B2[4])   verifier -> query chip
//   HINTR  (pointer, words)   ecall chip (a HINT_READ) -> hint chip, which puts the words it covers on the IMG bus
#include "machine_defs.hpp"

#include <cstdlib>
#include <mutex>

namespace zksp {
namespace mach {

namespace {

uint32_t mont(uint64_t canonical) { return Fp::from_canonical((uint32_t)(canonical % kP)).v; }

LinForm lf_zero() {
  LinForm f{};
  return f;
}
void lf_add(LinForm& f, int col, uint64_t coef) {
  if (f.n >= kLfMax) abort();
  f.col[f.n] = col;
  f.coef[f.n] = mont(coef);
  f.n++;
}
LinForm lf_col(int col) {
  LinForm f = lf_zero();
  lf_add(f, col, 1);
  return f;
}
LinForm lf_const(uint32_t c) {
  LinForm f = lf_zero();
  f.c0 = mont(c);
  return f;
}
LinForm lf_plus(LinForm f, uint32_t c) {
  f.c0 = (Fp::raw(f.c0) + Fp::raw(mont(c))).v;
  return f;
}
LinForm lf_pair(int a, int b, uint64_t k) {  // a + k * b
  LinForm f = lf_zero();
  lf_add(f, a, 1);
  lf_add(f, b, k);
  return f;
}
LinForm lf_sum(std::initializer_list<int> cols) {
  LinForm f = lf_zero();
  for (int c : cols) lf_add(f, c, 1);
  return f;
}
LinForm lf_const_minus(uint32_t c, int col) {  // c - col
  LinForm f = lf_zero();
  lf_add(f, col, kP - 1);
  f.c0 = mont(c);
  return f;
}
LinForm lf_bits(int bits, int n) {
  LinForm f = lf_zero();
  for (int i = 0; i < n; ++i) lf_add(f, bits + i, (uint64_t)1 << i);
  return f;
}
Interaction mem_inter(int sign, const LinForm& mult, const LinForm& addr, const LinForm& lo, const LinForm& hi, const LinForm& ts) {
  Interaction it{};
  it.bus = BUS_MEM; it.sign = sign; it.mult = mult; it.n_el = 4;
  it.el[0] = addr; it.el[1] = lo; it.el[2] = hi; it.el[3] = ts;
  return it;
}
Interaction range_inter(int sign, const LinForm& mult, const LinForm& kind, const LinForm& value) {
  Interaction it{};
  it.bus = BUS_RANGE; it.sign = sign; it.mult = mult; it.n_el = 2;
  it.el[0] = kind; it.el[1] = value;
  return it;
}
Interaction bytes_inter(int sign, const LinForm& mult, const LinForm& x, const LinForm& y) {
  Interaction it{};
  it.bus = BUS_BYTES; it.sign = sign; it.mult = mult; it.n_el = 2;
  it.el[0] = x; it.el[1] = y;
  return it;
}

constexpr int kCpuInter = 19;
Interaction g_cpu[kCpuInter], g_keccak[50], g_kmem[8], g_memfinal[10], g_image[1], g_program[1], g_mul[5], g_div[13], g_table[7], g_alu[1], g_sub[5], g_bw[5], g_p2[10], g_ecall[12], g_qr[17], g_tr[16], g_hint[2];
ChipDef g_chips[kNumChips];

void build() {
  const LinForm one = lf_const(1), zero = lf_const(0), ts = lf_col(C_TS);
  const LinForm a_lo = lf_col(C_A), a_hi = lf_col(C_A + 1), b_lo = lf_col(C_B), b_hi = lf_col(C_B + 1), c_lo = lf_col(C_C),
                c_hi = lf_col(C_C + 1);
  {
    Interaction& it = g_cpu[0];
    it = Interaction{};
    it.bus = BUS_PROG; it.sign = -1; it.mult = one; it.n_el = 13;
    it.el[0] = lf_col(C_PC);
    it.el[1] = lf_zero();
    for (int k = 1; k <= kNumCls; ++k) lf_add(it.el[1], selc(k), (uint64_t)k);
    it.el[2] = lf_col(C_CODE); it.el[3] = lf_col(C_UC); it.el[4] = lf_col(C_WR); it.el[5] = lf_col(C_USE2); it.el[6] = lf_col(C_RD);
    it.el[7] = lf_col(C_RS1); it.el[8] = lf_col(C_RS2); it.el[9] = lf_col(C_IMM_LO); it.el[10] = lf_col(C_IMM_HI);
    it.el[11] = lf_col(C_TGT_LO); it.el[12] = lf_col(C_TGT_HI);
  }
  // Three accesses per row: rs1 at ts, the second operand (rs2, or a load's memory word) at ts + 1, the written location
  // (rd, or a store's memory word) at ts + 2.  Previous access time of slot q: ts + q - 1 - (gap_lo + 2^16 gap_hi).
  LinForm pts[3];
  for (int q = 0; q < 3; ++q) {
    pts[q] = lf_zero();
    lf_add(pts[q], C_TS, 1); lf_add(pts[q], C_GAP + 2 * q, kP - 1); lf_add(pts[q], C_GAP + 2 * q + 1, kP - 65536);
    pts[q].c0 = mont((uint64_t)(kP + q - 1));
  }
  g_cpu[1] = mem_inter(-1, one, lf_col(C_RS1), b_lo, b_hi, pts[0]);
  g_cpu[2] = mem_inter(+1, one, lf_col(C_RS1), b_lo, b_hi, ts);
  {
    // rs2 (USE2) or a load: exclusive by the Program table (a load's second operand is its immediate)
    const LinForm m2 = lf_sum({C_USE2, selc(CL_LW), selc(CL_LDS)});
    g_cpu[3] = mem_inter(-1, m2, lf_col(C_ADDR2), c_lo, c_hi, pts[1]);
    g_cpu[4] = mem_inter(+1, m2, lf_col(C_ADDR2), c_lo, c_hi, lf_plus(ts, 1));
    // rd (WR) or a store (which writes no register)
    const LinForm m3 = lf_sum({C_WR, selc(CL_SW), selc(CL_STS)});
    g_cpu[5] = mem_inter(-1, m3, lf_col(C_ADDR3), lf_col(C_W_PLO), lf_col(C_W_PHI), pts[2]);
    g_cpu[6] = mem_inter(+1, m3, lf_col(C_ADDR3), a_lo, a_hi, lf_plus(ts, 2));
  }
  // access-time differences: every row looks up its three low limbs and the high bytes
  for (int q = 0; q < 3; ++q) g_cpu[7 + q] = range_inter(-1, one, zero, lf_col(C_GAP + 2 * q));
  g_cpu[10] = bytes_inter(-1, one, lf_col(C_GAP + 1), lf_col(C_GAP + 3));
  g_cpu[11] = bytes_inter(-1, one, lf_col(C_GAP + 5), zero);
  {
    // the adder output is canonical, an address is word-aligned once its byte offset is taken off, and addresses,
    // jump targets and the keccak call's return address stay below 0x78000000
    // ... and so is the difference of an unsigned comparison (UC)
    const LinForm chk = lf_sum({selc(CL_ADD), selc(CL_SUB), selc(CL_JALR), selc(CL_LW), selc(CL_SW), selc(CL_LDS), selc(CL_STS),
                                selc(CL_ECALL), selc(CL_KECCAK), C_UC});
    LinForm xoff = lf_col(C_X);
    lf_add(xoff, C_O1, kP - 1); lf_add(xoff, C_O2, kP - 2); lf_add(xoff, C_O3, kP - 3);
    LinForm top = lf_zero();  // kind 2: the high limb of an address is in 1 .. kAddrHiMax (0x10000 <= address < 0x77FF0000)
    for (int c : {selc(CL_JALR), selc(CL_LW), selc(CL_SW), selc(CL_LDS), selc(CL_STS), selc(CL_KECCAK)}) lf_add(top, c, 2);
    g_cpu[12] = range_inter(-1, chk, top, lf_col(C_X + 1));
    g_cpu[13] = range_inter(-1, chk, lf_sum({selc(CL_JALR), selc(CL_LW), selc(CL_SW), selc(CL_LDS), selc(CL_STS)}), xoff);
  }
  {
    Interaction& al = g_cpu[14];
    al = Interaction{};
    // every ALU-class instruction and every ordered branch, except the unsigned comparisons the row does itself
    al.bus = BUS_ALU; al.sign = +1; al.mult = lf_sum({selc(CL_ALU), selc(CL_BLT), selc(CL_BGE)}); lf_add(al.mult, C_UC, kP - 1); al.n_el = 7;
    al.el[0] = lf_col(C_CODE);
    al.el[1] = a_lo; al.el[2] = a_hi; al.el[3] = b_lo; al.el[4] = b_hi; al.el[5] = c_lo; al.el[6] = c_hi;
    // sub-word loads: (op, offset, value loaded, memory word, 0); sub-word stores: (op, offset, word left behind, word
    // before, low limb of the stored register) - one tuple layout, two sends because the word sits in other columns
    for (int st = 0; st < 2; ++st) {
      Interaction& sb = g_cpu[15 + st];
      sb = Interaction{};
      sb.bus = BUS_SUB; sb.sign = +1; sb.mult = lf_col(selc(st ? CL_STS : CL_LDS)); sb.n_el = 7;
      sb.el[0] = lf_col(C_CODE);
      sb.el[1] = lf_zero(); lf_add(sb.el[1], C_O1, 1); lf_add(sb.el[1], C_O2, 2); lf_add(sb.el[1], C_O3, 3);
      sb.el[2] = a_lo; sb.el[3] = a_hi;
      sb.el[4] = st ? lf_col(C_W_PLO) : c_lo; sb.el[5] = st ? lf_col(C_W_PHI) : c_hi;
      sb.el[6] = st ? c_lo : zero;
    }
    Interaction& kc = g_cpu[17];
    kc = Interaction{};
    kc.bus = BUS_KCALL; kc.sign = +1; kc.mult = lf_col(selc(CL_KECCAK)); kc.n_el = 3;
    kc.el[0] = ts; kc.el[1] = c_lo; kc.el[2] = c_hi;
    // an ecall: the ecall chip takes it from here (time, pc, next pc, the code in t0, the value left in t0)
    Interaction& ec = g_cpu[18];
    ec = Interaction{};
    ec.bus = BUS_ECALL; ec.sign = +1; ec.mult = lf_col(selc(CL_ECALL)); ec.n_el = 6;
    ec.el[0] = ts; ec.el[1] = lf_col(C_PC); ec.el[2] = lf_col(C_NEXT_PC); ec.el[3] = b_lo; ec.el[4] = a_lo; ec.el[5] = a_hi;
  }
  {
    // ecall chip: receives the CPU row's hand-over, reads a0 (at ts + 1) and a1 (at ts + 2) itself, sends the COMMIT /
    // COMMIT_DEFERRED words and HALT's exit code to the buses the verifier closes
    const LinForm real = lf_col(EC_IS_REAL), ets = lf_col(EC_TS);
    const LinForm ec_lo = lf_col(EC_C_LO), ec_hi = lf_col(EC_C_HI), em_lo = lf_col(EC_M_LO), em_hi = lf_col(EC_M_HI);
    Interaction& rc = g_ecall[0];
    rc = Interaction{};
    rc.bus = BUS_ECALL; rc.sign = -1; rc.mult = real; rc.n_el = 6;
    rc.el[0] = ets; rc.el[1] = lf_col(EC_PC); rc.el[2] = lf_col(EC_NP); rc.el[3] = lf_col(EC_B_LO);
    rc.el[4] = lf_col(EC_A_LO); rc.el[5] = lf_col(EC_A_HI);
    LinForm epts[2];
    for (int q = 0; q < 2; ++q) {
      epts[q] = lf_zero();
      lf_add(epts[q], EC_TS, 1); lf_add(epts[q], EC_GAP + 2 * q, kP - 1); lf_add(epts[q], EC_GAP + 2 * q + 1, kP - 65536);
      epts[q].c0 = mont((uint64_t)q);
    }
    g_ecall[1] = mem_inter(-1, real, lf_const(10), ec_lo, ec_hi, epts[0]);
    g_ecall[2] = mem_inter(+1, real, lf_const(10), ec_lo, ec_hi, lf_plus(ets, 1));
    g_ecall[3] = mem_inter(-1, real, lf_const(11), em_lo, em_hi, epts[1]);
    g_ecall[4] = mem_inter(+1, real, lf_const(11), em_lo, em_hi, lf_plus(ets, 2));
    g_ecall[5] = range_inter(-1, real, lf_const(0), lf_col(EC_GAP));
    g_ecall[6] = range_inter(-1, real, lf_const(0), lf_col(EC_GAP + 2));
    g_ecall[7] = bytes_inter(-1, real, lf_col(EC_GAP + 1), lf_col(EC_GAP + 3));
    Interaction& pc = g_ecall[8];
    pc = Interaction{};
    pc.bus = BUS_PUBC; pc.sign = +1; pc.n_el = 4;
    pc.mult = lf_pair(EC_SC + SC_COMMIT, EC_SC + SC_DEFER, 1);
    pc.el[0] = lf_pair(EC_SC + SC_COMMIT, EC_SC + SC_DEFER, 2);
    pc.el[1] = ec_lo; pc.el[2] = em_lo; pc.el[3] = em_hi;
    Interaction& ph = g_ecall[9];
    ph = Interaction{};
    ph.bus = BUS_PUBH; ph.sign = +1; ph.mult = lf_col(EC_SC + SC_HALT); ph.n_el = 2;
    ph.el[0] = ec_lo; ph.el[1] = ec_hi;
    // a HINT_READ announces (pointer, words) to the hint chip; the word count is a 16-bit value
    Interaction& hr = g_ecall[10];
    hr = Interaction{};
    hr.bus = BUS_HINTR; hr.sign = +1; hr.mult = lf_col(EC_SC + SC_HINT_READ); hr.n_el = 2;
    hr.el[0] = lf_pair(EC_C_LO, EC_C_HI, 65536); hr.el[1] = lf_col(EC_NW);
    g_ecall[11] = range_inter(-1, lf_col(EC_SC + SC_HINT_READ), lf_const(0), lf_col(EC_NW));
  }
  {
    // hint chip: a read's first word takes the announcement; a word the run touches hands its initial value to the memory
    // boundary chip over the IMG bus
    Interaction& rc = g_hint[0];
    rc = Interaction{};
    rc.bus = BUS_HINTR; rc.sign = -1; rc.mult = lf_col(HN_FIRST); rc.n_el = 2;
    rc.el[0] = lf_col(HN_ADDR); rc.el[1] = lf_col(HN_CNT);
    Interaction& im = g_hint[1];
    im = Interaction{};
    im.bus = BUS_IMG; im.sign = +1; im.mult = lf_col(HN_USED); im.n_el = 3;
    im.el[0] = lf_col(HN_ADDR); im.el[1] = lf_col(HN_LO); im.el[2] = lf_col(HN_HI);
  }
  g_chips[kHint] = {"hint", 0, kHintWidth, 2, g_hint, kHintConstraints, 0};
  for (int i = 0; i < 50; ++i) {
    Interaction& it = g_keccak[i];
    it = Interaction{};
    const int lane = i >> 1, half = i & 1;
    const int out = lane == 0 ? ka::kAppp00 + 2 * half : ka::kApp + 4 * lane + 2 * half;
    it.bus = BUS_KIO; it.sign = +1; it.mult = lf_col(ka::kExport); it.n_el = 6;
    it.el[0] = lf_col(KC_TS); it.el[1] = lf_const((uint32_t)i);
    it.el[2] = lf_col(ka::kPreimage + 4 * lane + 2 * half); it.el[3] = lf_col(ka::kPreimage + 4 * lane + 2 * half + 1);
    it.el[4] = lf_col(out); it.el[5] = lf_col(out + 1);
  }
  {
    Interaction& kc = g_kmem[0];
    kc = Interaction{};
    kc.bus = BUS_KCALL; kc.sign = -1; kc.mult = lf_col(KM_CALL); kc.n_el = 3;
    kc.el[0] = lf_col(KM_TS); kc.el[1] = lf_col(KM_PTR_LO); kc.el[2] = lf_col(KM_PTR_HI);
    Interaction& io = g_kmem[1];
    io = Interaction{};
    io.bus = BUS_KIO; io.sign = -1; io.mult = lf_col(KM_IS_REAL); io.n_el = 6;
    io.el[0] = lf_col(KM_TS); io.el[1] = lf_col(KM_IDX); io.el[2] = lf_col(KM_OLD_LO); io.el[3] = lf_col(KM_OLD_HI);
    io.el[4] = lf_col(KM_NEW_LO); io.el[5] = lf_col(KM_NEW_HI);
    g_kmem[2] = mem_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_OLD_LO), lf_col(KM_OLD_HI), lf_col(KM_PTS));
    g_kmem[3] = mem_inter(+1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_NEW_LO), lf_col(KM_NEW_HI), lf_plus(lf_col(KM_TS), 2));
    g_kmem[4] = range_inter(-1, lf_col(KM_IS_REAL), zero, lf_col(KM_GL));
    g_kmem[5] = bytes_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_GH), zero);
    // the state pointer is word-aligned, lies at 0x10000 or above (not in register space) and the 200 bytes end below
    // 0x78000000: its high limb is looked up as kind 2 (1 .. kAddrHiMax = 0x77FE)
    g_kmem[6] = range_inter(-1, lf_col(KM_CALL), one, lf_col(KM_PTR_LO));
    g_kmem[7] = range_inter(-1, lf_col(KM_CALL), lf_const(2), lf_col(KM_PTR_HI));
  }
  {
    const LinForm real = lf_col(MF_IS_REAL), init = lf_col(MF_IS_INIT), addr = lf_pair(MF_LO, MF_HI, 65536);
    g_memfinal[0] = mem_inter(-1, real, addr, lf_col(MF_FIN_LO), lf_col(MF_FIN_HI), lf_col(MF_FIN_TS));
    g_memfinal[1] = mem_inter(+1, real, addr, lf_col(MF_INIT_LO), lf_col(MF_INIT_HI), zero);
    Interaction& im = g_memfinal[2];
    im = Interaction{};
    im.bus = BUS_IMG; im.sign = -1; im.mult = lf_pair(MF_IS_REAL, MF_IS_ZERO, kP - 1); im.n_el = 3;  // an image word (image chip) or a hinted one (hint chip)
    im.el[0] = addr; im.el[1] = lf_col(MF_INIT_LO); im.el[2] = lf_col(MF_INIT_HI);
    g_memfinal[3] = range_inter(-1, real, zero, lf_col(MF_LO));
    g_memfinal[4] = range_inter(-1, real, zero, lf_col(MF_HI));
    g_memfinal[5] = range_inter(-1, real, zero, lf_const_minus(kAddrHiMax, MF_HI));
    g_memfinal[6] = range_inter(-1, real, zero, lf_col(MF_D_LO));
    g_memfinal[7] = range_inter(-1, real, zero, lf_col(MF_D_HI));
    g_memfinal[8] = range_inter(-1, init, zero, lf_col(MF_INIT_LO));
    g_memfinal[9] = range_inter(-1, init, zero, lf_col(MF_INIT_HI));
  }
  {
    Interaction& im = g_image[0];
    im = Interaction{};
    im.bus = BUS_IMG; im.sign = +1; im.mult = lf_col(kImagePrepWidth + 0); im.n_el = 3;
    im.el[0] = lf_col(IMG_P_ADDR); im.el[1] = lf_col(IMG_P_LO); im.el[2] = lf_col(IMG_P_HI);
    Interaction& pr = g_program[0];
    pr = Interaction{};
    pr.bus = BUS_PROG; pr.sign = +1; pr.mult = lf_col(kProgramPrepWidth + 0); pr.n_el = 13;
    for (int j = 0; j < 13; ++j) pr.el[j] = lf_col(j);
    const LinForm idx = lf_pair(TB_P_X, TB_P_Y, 256);
    g_table[0] = range_inter(+1, lf_col(kTablePrepWidth + TB_M_R16), zero, idx);
    g_table[1] = range_inter(+1, lf_col(kTablePrepWidth + TB_M_AL), one, idx);
    g_table[2] = range_inter(+1, lf_col(kTablePrepWidth + TB_M_TOP), lf_const(2), idx);
    g_table[3] = bytes_inter(+1, lf_col(kTablePrepWidth + TB_M_BY), lf_col(TB_P_X), lf_col(TB_P_Y));
    // byte operations (kind, x, y, z): 1 xor, 2 or (= x + y - and), 3 and
    for (int k = 0; k < 3; ++k) {
      Interaction& it = g_table[4 + k];
      it = Interaction{};
      it.bus = BUS_BYTEOP; it.sign = +1; it.mult = lf_col(kTablePrepWidth + TB_M_XOR + k); it.n_el = 4;
      it.el[0] = lf_const((uint32_t)k + 1); it.el[1] = lf_col(TB_P_X); it.el[2] = lf_col(TB_P_Y);
      if (k == 0) it.el[3] = lf_col(TB_P_XOR);
      else if (k == 2) it.el[3] = lf_col(TB_P_AND);
      else { it.el[3] = lf_pair(TB_P_X, TB_P_Y, 1); lf_add(it.el[3], TB_P_AND, kP - 1); }
    }
  }
  for (int hi = 0; hi < 2; ++hi) {
    Interaction& it = g_mul[hi];
    it = Interaction{};
    it.bus = BUS_ALU; it.sign = -1; it.n_el = 7;
    it.mult = hi ? lf_col(MU_HI) : lf_pair(MU_IS_REAL, MU_HI, kP - 1);
    if (!hi) { lf_add(it.mult, MU_SH, kP - 1); lf_add(it.mult, MU_SHU, kP - 1); }  // mul: the rows that are none of the high words
    it.el[0] = lf_const(hi ? (uint32_t)MULHU : (uint32_t)MUL);
    it.el[1] = lf_bits(MU_P + 32 * hi, 16); it.el[2] = lf_bits(MU_P + 32 * hi + 16, 16);
    it.el[3] = lf_bits(MU_B, 16); it.el[4] = lf_bits(MU_B + 16, 16); it.el[5] = lf_bits(MU_C, 16); it.el[6] = lf_bits(MU_C + 16, 16);
  }
  {
    // mulh / mulhsu: the signed high word R, range-checked
    Interaction& it = g_mul[2];
    it = Interaction{};
    it.bus = BUS_ALU; it.sign = -1; it.n_el = 7;
    it.mult = lf_pair(MU_SH, MU_SHU, 1);
    it.el[0] = lf_zero(); lf_add(it.el[0], MU_SH, MULH); lf_add(it.el[0], MU_SHU, MULHSU);
    it.el[1] = lf_col(MU_R); it.el[2] = lf_col(MU_R + 1);
    it.el[3] = lf_bits(MU_B, 16); it.el[4] = lf_bits(MU_B + 16, 16); it.el[5] = lf_bits(MU_C, 16); it.el[6] = lf_bits(MU_C + 16, 16);
    g_mul[3] = range_inter(-1, lf_pair(MU_SH, MU_SHU, 1), zero, lf_col(MU_R));
    g_mul[4] = range_inter(-1, lf_pair(MU_SH, MU_SHU, 1), zero, lf_col(MU_R + 1));
  }
  {
    // divider chip (air_machine.hpp): the instruction from the CPU row, the product |q| |d| from the multiplier chip (low
    // word PL, high word zero), range lookups
    const LinForm real = lf_col(DV_IS_REAL), nzd = lf_col(DV_NZD), sgn = lf_pair(DV_F + 0, DV_F + 2, 1);
    Interaction& rc = g_div[0];
    rc = Interaction{};
    rc.bus = BUS_ALU; rc.sign = -1; rc.mult = real; rc.n_el = 7;
    rc.el[0] = lf_zero();
    lf_add(rc.el[0], DV_F + 0, DIV); lf_add(rc.el[0], DV_F + 1, DIVU); lf_add(rc.el[0], DV_F + 2, REM); lf_add(rc.el[0], DV_F + 3, REMU);
    rc.el[1] = lf_col(DV_A); rc.el[2] = lf_col(DV_A + 1); rc.el[3] = lf_col(DV_N); rc.el[4] = lf_col(DV_N + 1);
    rc.el[5] = lf_col(DV_D); rc.el[6] = lf_col(DV_D + 1);
    for (int hi = 0; hi < 2; ++hi) {
      Interaction& ml = g_div[1 + hi];
      ml = Interaction{};
      ml.bus = BUS_ALU; ml.sign = +1; ml.mult = nzd; ml.n_el = 7;
      ml.el[0] = lf_const(hi ? (uint32_t)MULHU : (uint32_t)MUL);
      ml.el[1] = hi ? zero : lf_col(DV_PL); ml.el[2] = hi ? zero : lf_col(DV_PL + 1);
      ml.el[3] = lf_col(DV_AQ); ml.el[4] = lf_col(DV_AQ + 1); ml.el[5] = lf_col(DV_AD); ml.el[6] = lf_col(DV_AD + 1);
    }
    const int checked[8] = {DV_A, DV_A + 1, DV_AN, DV_AN + 1, DV_AR, DV_AR + 1, DV_E, DV_E + 1};
    for (int k = 0; k < 8; ++k) g_div[3 + k] = range_inter(-1, real, zero, lf_col(checked[k]));
    LinForm nh2 = lf_zero(), dh2 = lf_zero();
    lf_add(nh2, DV_NH, 2); lf_add(dh2, DV_DH, 2);
    g_div[11] = range_inter(-1, sgn, zero, nh2);
    g_div[12] = range_inter(-1, sgn, zero, dh2);
  }
  g_chips[kDiv] = {"divider", 0, kDivWidth, 13, g_div, kDivConstraints, 0};
  {
    Interaction& it = g_alu[0];
    it = Interaction{};
    it.bus = BUS_ALU; it.sign = -1; it.mult = lf_col(AL_IS_REAL); it.n_el = 7;
    it.el[0] = lf_zero();
    for (int k = 0; k < 4; ++k) lf_add(it.el[0], AL_SEL + k, (uint64_t)(SLL + k));
    it.el[1] = lf_col(AL_A); it.el[2] = lf_col(AL_A + 1);
    it.el[3] = lf_bits(AL_B, 16); it.el[4] = lf_bits(AL_B + 16, 16); it.el[5] = lf_bits(AL_C, 16); it.el[6] = lf_bits(AL_C + 16, 16);
  }
  {
    // bitwise: the word tuple from the CPU row, and one byte-operation lookup per byte
    Interaction& it = g_bw[0];
    it = Interaction{};
    it.bus = BUS_ALU; it.sign = -1; it.mult = lf_col(BW_IS_REAL); it.n_el = 7;
    it.el[0] = lf_zero();
    for (int k = 0; k < 3; ++k) lf_add(it.el[0], BW_SEL + k, (uint64_t)(XOR + k));
    for (int w = 0; w < 3; ++w) {  // a, b, c as limbs of two bytes
      const int base = w == 0 ? BW_A : w == 1 ? BW_B : BW_C;
      it.el[1 + 2 * w] = lf_pair(base, base + 1, 256);
      it.el[2 + 2 * w] = lf_pair(base + 2, base + 3, 256);
    }
    for (int i = 0; i < 4; ++i) {
      Interaction& op = g_bw[1 + i];
      op = Interaction{};
      op.bus = BUS_BYTEOP; op.sign = -1; op.mult = lf_col(BW_IS_REAL); op.n_el = 4;
      op.el[0] = lf_zero();
      for (int k = 0; k < 3; ++k) lf_add(op.el[0], BW_SEL + k, (uint64_t)k + 1);
      op.el[1] = lf_col(BW_B + i); op.el[2] = lf_col(BW_C + i); op.el[3] = lf_col(BW_A + i);
    }
  }
  {
    static const uint32_t codes[6] = {LB, LH, LBU, LHU, SB, SH};
    Interaction& it = g_sub[0];
    it = Interaction{};
    it.bus = BUS_SUB; it.sign = -1; it.mult = lf_col(SW_IS_REAL); it.n_el = 7;
    it.el[0] = lf_zero();
    for (int k = 0; k < 6; ++k) lf_add(it.el[0], SW_SEL + k, codes[k]);
    it.el[1] = lf_zero(); lf_add(it.el[1], SW_O + 1, 1); lf_add(it.el[1], SW_O + 2, 2); lf_add(it.el[1], SW_O + 3, 3);
    it.el[2] = lf_col(SW_A); it.el[3] = lf_col(SW_A + 1);
    it.el[4] = lf_pair(SW_MB, SW_MB + 1, 256); it.el[5] = lf_pair(SW_MB + 2, SW_MB + 3, 256); it.el[6] = lf_pair(SW_CB, SW_CB + 1, 256);
    // the bytes are bytes; the sign bit of a signed load is bit 7 of the byte it extends (byte AND 0x80 = 128 * sign)
    g_sub[1] = bytes_inter(-1, lf_col(SW_IS_REAL), lf_col(SW_MB), lf_col(SW_MB + 1));
    g_sub[2] = bytes_inter(-1, lf_col(SW_IS_REAL), lf_col(SW_MB + 2), lf_col(SW_MB + 3));
    g_sub[3] = bytes_inter(-1, lf_col(SW_IS_REAL), lf_col(SW_CB), lf_col(SW_CB + 1));
    Interaction& sg = g_sub[4];
    sg = Interaction{};
    sg.bus = BUS_BYTEOP; sg.sign = -1; sg.mult = lf_pair(SW_SEL + 0, SW_SEL + 1, 1); sg.n_el = 4;
    sg.el[0] = lf_const(3);  // and
    sg.el[1] = lf_col(SW_SELB); sg.el[2] = lf_const(0x80);
    sg.el[3] = lf_zero(); lf_add(sg.el[3], SW_S, 128);
  }
  {
    // Poseidon2 (air_machine.hpp "Poseidon2 chip").  DIGEST tuples are (tag, type, key, mask, 8 words); type 2: a heap
    // node (stage 1: children in, parent out), type 1: the hash of an injected matrix row (a sponge's last row -> the
    // injection row with the same labels), type 0: the end of a run (-> the verifier, who knows the root).  The output
    // words are the external linear layer applied to the last round's S-box outputs (circ(2 M4, M4, M4, M4),
    // M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]).
    static const uint32_t m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    const int ylast = P2_EXT + 32 * 7 + 16;
    const LinForm tag = lf_col(P2_T), mask = lf_col(P2_M), key = lf_pair(P2_KL, P2_KH, 65536);
    for (int side = 0; side < 2; ++side) {
      Interaction& it = g_p2[side];
      it = Interaction{};
      it.bus = BUS_DIGEST; it.sign = -1; it.mult = lf_col(P2_FN); it.n_el = 12;
      it.el[0] = tag; it.el[1] = lf_const(2);
      it.el[2] = lf_zero(); lf_add(it.el[2], P2_KL, 2); lf_add(it.el[2], P2_KH, 2 * 65536); it.el[2].c0 = mont((uint64_t)side);
      it.el[3] = mask;
      for (int j = 0; j < 8; ++j) it.el[4 + j] = lf_col(P2_IN + 8 * side + j);
    }
    Interaction& inj = g_p2[2];
    inj = Interaction{};
    inj.bus = BUS_DIGEST; inj.sign = -1; inj.mult = lf_col(P2_FJ); inj.n_el = 12;
    inj.el[0] = tag; inj.el[1] = lf_const(1); inj.el[2] = key; inj.el[3] = mask;
    for (int j = 0; j < 8; ++j) inj.el[4 + j] = lf_col(P2_IN + 8 + j);
    Interaction& out = g_p2[3];
    out = Interaction{};
    out.bus = BUS_DIGEST; out.sign = +1; out.mult = lf_pair(P2_FN, P2_SND, 1); out.n_el = 12;
    out.el[0] = tag;
    out.el[1] = lf_zero(); lf_add(out.el[1], P2_FN, 2); lf_add(out.el[1], P2_SZ, 1); lf_add(out.el[1], P2_SC, 1);
    out.el[2] = key; out.el[3] = mask;
    for (int j = 0; j < 8; ++j) {
      out.el[4 + j] = lf_zero();
      for (int i = 0; i < 16; ++i) lf_add(out.el[4 + j], ylast + i, (uint64_t)m4[j & 3][i & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u));
    }
    // the pair a FRI leaf hashes goes to the fold chip
    Interaction& pair = g_p2[4];
    pair = Interaction{};
    pair.bus = BUS_PAIR; pair.sign = +1; pair.mult = lf_col(P2_FR); pair.n_el = 9;
    pair.el[0] = tag;
    for (int j = 0; j < 8; ++j) pair.el[1 + j] = lf_col(P2_IN + j);
  }
  // the key's limbs: 16 bits, and twice the high limb plus one in 1 .. kAddrHiMax (kind 2): the key stays below 0x3C000000
  // and its children's keys 2K, 2K + 1 below 0x78000000 < p - no key aliases another mod p
  g_p2[5] = range_inter(-1, lf_col(P2_IS_REAL), lf_const(0), lf_col(P2_KL));
  {
    LinForm kh2 = lf_zero();
    lf_add(kh2, P2_KH, 2);
    kh2.c0 = mont(1);
    g_p2[6] = range_inter(-1, lf_col(P2_IS_REAL), lf_const(2), kh2);
  }
  {
    // stage 2b: the end of a run tells the query chip where it sits and takes the root from the transcript chip; the end of a
    // matrix row's hash hands over its Horner sum
    static const uint32_t m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    const int ylast = P2_EXT + 32 * 7 + 16;
    const LinForm tag = lf_col(P2_T), mask = lf_col(P2_M), key = lf_pair(P2_KL, P2_KH, 65536);
    Interaction& pos = g_p2[7];
    pos = Interaction{};
    pos.bus = BUS_POS; pos.sign = +1; pos.mult = lf_col(P2_RE); pos.n_el = 4;
    pos.el[0] = tag; pos.el[1] = key; pos.el[2] = mask; pos.el[3] = lf_col(P2_RID);
    Interaction& rt = g_p2[8];
    rt = Interaction{};
    rt.bus = BUS_ROOT; rt.sign = -1; rt.mult = lf_col(P2_RE); rt.n_el = 9;
    rt.el[0] = lf_col(P2_RID);
    for (int j = 0; j < 8; ++j) {
      rt.el[1 + j] = lf_zero();
      for (int i = 0; i < 16; ++i) lf_add(rt.el[1 + j], ylast + i, (uint64_t)m4[j & 3][i & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u));
    }
    Interaction& sg = g_p2[9];
    sg = Interaction{};
    sg.bus = BUS_SEG; sg.sign = +1; sg.mult = lf_col(P2_SE); sg.n_el = 11;
    sg.el[0] = tag; sg.el[1] = key; sg.el[2] = mask;
    for (int j = 0; j < 4; ++j) { sg.el[3 + j] = lf_col(P2_SO + j); sg.el[7 + j] = lf_col(P2_AP + j); }
  }
  {
    // query chip (air_machine.hpp).  T(r) = 1 + 64 QL + 2^18 LEAF + r, RID(r) = 64 LEAF + r.
    auto tag_of = [&](uint32_t r, int kcol) {
      LinForm f = lf_zero();
      lf_add(f, QR_QL, kLeafTagStride); lf_add(f, QR_LEAF, kLeafTagLeafStride);
      if (kcol >= 0) lf_add(f, kcol, 1);
      f.c0 = mont(1 + r);
      return f;
    };
    auto rid_of = [&](uint32_t r, int kcol) {
      LinForm f = lf_zero();
      lf_add(f, QR_LEAF, 64);
      if (kcol >= 0) lf_add(f, kcol, 1);
      f.c0 = mont(r);
      return f;
    };
    const LinForm leaf = lf_col(QR_LEAF), last = lf_col(QR_LAST), lay = lf_col(QR_LAY), hasro = lf_col(QR_HASRO);
    int n = 0;
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_QIDX; it.sign = -1; it.mult = last; it.n_el = 3;
      it.el[0] = leaf; it.el[1] = lf_col(QR_QL); it.el[2] = lf_col(QR_ACC);
    }
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_LEAFK; it.sign = -1; it.mult = last; it.n_el = 3;
      it.el[0] = leaf; it.el[1] = lf_col(QR_K); it.el[2] = lf_col(QR_OMI);
    }
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_FINAL; it.sign = -1; it.mult = last; it.n_el = 5;
      it.el[0] = leaf;
      for (int j = 0; j < 4; ++j) it.el[1 + j] = lf_col(QR_F + j);
    }
    for (uint32_t r = 1; r <= 3; ++r) {  // the three trees as tall as the proof: position 2^(lm + 1) + 2 m + cs on the coset bit's row
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_POS; it.sign = -1; it.mult = lf_col(QR_CSR); it.n_el = 4;
      it.el[0] = tag_of(r, -1);
      it.el[1] = lf_zero(); lf_add(it.el[1], QR_POW, 2); lf_add(it.el[1], QR_LOW, 2); lf_add(it.el[1], QR_BIT, 1);
      it.el[2] = lf_col(QR_MT); it.el[3] = rid_of(r, -1);
    }
    {
      Interaction& it = g_qr[n++];  // the preprocessed tree (2^16 tall)
      it = Interaction{};
      it.bus = BUS_POS; it.sign = -1; it.mult = lf_col(QR_PR0); it.n_el = 4;
      it.el[0] = tag_of(0, -1);
      it.el[1] = lf_zero(); lf_add(it.el[1], QR_POW, 2); lf_add(it.el[1], QR_LOW, 2); lf_add(it.el[1], QR_CS, 1);
      it.el[2] = lf_col(QR_MT0); it.el[3] = rid_of(0, -1);
    }
    {
      Interaction& it = g_qr[n++];  // the FRI layer's tree: 2^(J + 1) + 2 rev(m mod 2^J) + cs
      it = Interaction{};
      it.bus = BUS_POS; it.sign = -1; it.mult = lay; it.n_el = 4;
      it.el[0] = tag_of(4, QR_K);
      it.el[1] = lf_zero(); lf_add(it.el[1], QR_POW, 2); lf_add(it.el[1], QR_REV, 2); lf_add(it.el[1], QR_CS, 1);
      it.el[2] = lf_const(0); it.el[3] = rid_of(4, QR_K);
    }
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_PAIR; it.sign = -1; it.mult = lay; it.n_el = 9;
      it.el[0] = tag_of(4, QR_K);
      for (int j = 0; j < 4; ++j) { it.el[1 + j] = lf_col(QR_LO + j); it.el[5 + j] = lf_col(QR_HI + j); }
    }
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_BETA; it.sign = -1; it.mult = lay; it.n_el = 6;
      it.el[0] = leaf; it.el[1] = lf_col(QR_K);
      for (int j = 0; j < 4; ++j) it.el[2 + j] = lf_col(QR_BETA + j);
    }
    for (uint32_t r = 0; r <= 3; ++r) {  // the Horner sums of the opened rows of the height that joins here
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_SEG; it.sign = -1; it.mult = r == 0 ? lf_col(QR_HAS0) : hasro; it.n_el = 11;
      it.el[0] = tag_of(r, -1); it.el[1] = lf_col(r == 0 ? QR_KEY0 : QR_KEYJ); it.el[2] = lf_col(r == 0 ? QR_M0 : QR_MJ);
      for (int j = 0; j < 4; ++j) { it.el[3 + j] = lf_col(QR_H + 4 * (int)r + j); it.el[7 + j] = lf_col(QR_AF + j); }
    }
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_ZETA; it.sign = -1; it.mult = hasro; it.n_el = 5;
      it.el[0] = leaf;
      for (int j = 0; j < 4; ++j) it.el[1 + j] = lf_col(QR_ZETA + j);
    }
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_AF; it.sign = -1; it.mult = hasro; it.n_el = 9;
      it.el[0] = leaf;
      for (int j = 0; j < 4; ++j) { it.el[1 + j] = lf_col(QR_AF + j); it.el[5 + j] = lf_col(QR_DL + j); }
    }
    {
      Interaction& it = g_qr[n++];
      it = Interaction{};
      it.bus = BUS_BCONST; it.sign = -1; it.mult = hasro; it.n_el = 12;
      it.el[0] = leaf; it.el[1] = lf_col(QR_K); it.el[2] = lf_col(QR_HAS0); it.el[3] = lf_col(QR_WH);
      for (int j = 0; j < 4; ++j) { it.el[4 + j] = lf_col(QR_B1 + j); it.el[8 + j] = lf_col(QR_B2 + j); }
    }
    if (n != 17) abort();
  }
  {
    // transcript chip (air_machine.hpp).  The output words of a row: the external linear layer of the last round's columns.
    static const uint32_t m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    const int ylast = TR_EXT + 32 * 7 + 16;
    auto out_word = [&](int j) {
      LinForm f = lf_zero();
      for (int i = 0; i < 16; ++i) lf_add(f, ylast + i, (uint64_t)m4[j & 3][i & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u));
      return f;
    };
    const LinForm leaf = lf_col(TR_LEAF), step = lf_col(TR_STEP);
    LinForm flags = lf_zero();
    for (int k = 0; k < 7; ++k) lf_add(flags, TR_UROOT + k, (uint64_t)1 << k);
    for (int k = 0; k < 8; ++k) lf_add(flags, TR_QM + k, (uint64_t)128 << k);
    int n = 0;
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_TBLK; it.sign = -1; it.mult = lf_col(TR_ABS); it.n_el = 12;
      it.el[0] = leaf; it.el[1] = step; it.el[2] = flags; it.el[3] = lf_col(TR_RIDK);
      for (int j = 0; j < 8; ++j) it.el[4 + j] = lf_col(TR_IN + j);
    }
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_TSQ; it.sign = -1; it.mult = lf_pair(TR_IS_REAL, TR_ABS, kP - 1); it.n_el = 4;
      it.el[0] = leaf; it.el[1] = step; it.el[2] = flags; it.el[3] = lf_col(TR_QBASE);
    }
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_ROOT; it.sign = +1; it.mult = lf_col(TR_MROOT); it.n_el = 9;
      it.el[0] = lf_zero(); lf_add(it.el[0], TR_LEAF, 64); lf_add(it.el[0], TR_RIDK, 1);
      for (int j = 0; j < 8; ++j) it.el[1 + j] = lf_col(TR_IN + j);
    }
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_FINAL; it.sign = +1; it.mult = lf_col(TR_MFIN); it.n_el = 5;
      it.el[0] = leaf;
      for (int j = 0; j < 4; ++j) it.el[1 + j] = lf_col(TR_IN + j);
    }
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_ZETA; it.sign = +1; it.mult = lf_col(TR_MZETA); it.n_el = 5;
      it.el[0] = leaf;
      for (int j = 0; j < 4; ++j) it.el[1 + j] = out_word(7 - j);
    }
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_AF; it.sign = +1; it.mult = lf_col(TR_MAF); it.n_el = 9;
      it.el[0] = leaf;
      for (int j = 0; j < 8; ++j) it.el[1 + j] = out_word(7 - j);
    }
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_BETA; it.sign = +1; it.mult = lf_col(TR_MBETA); it.n_el = 6;
      it.el[0] = leaf; it.el[1] = lf_plus(lf_col(TR_RIDK), kP - 4);
      for (int j = 0; j < 4; ++j) it.el[2 + j] = out_word(7 - j);
    }
    {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_POW; it.sign = +1; it.mult = lf_col(TR_UPOW); it.n_el = 2;
      it.el[0] = leaf; it.el[1] = out_word(7);
    }
    for (int j = 0; j < 8; ++j) {
      Interaction& it = g_tr[n++];
      it = Interaction{};
      it.bus = BUS_QIDX; it.sign = +1; it.mult = lf_col(TR_QM + j); it.n_el = 3;
      it.el[0] = leaf; it.el[1] = lf_plus(lf_col(TR_QBASE), (uint32_t)j); it.el[2] = out_word(7 - j);
    }
    if (n != 16) abort();
  }
  g_chips[kQr] = {"query", 0, kQrWidth, 17, g_qr, kQrConstraints, 0};
  g_chips[kTr] = {"transcript", 0, kTrWidth, 16, g_tr, kTrConstraints, 0};
  g_chips[kP2] = {"poseidon2", 0, kP2Width, 10, g_p2, kP2Constraints, 0};
  g_chips[kTable] = {"table", kTablePrepWidth, kTableWidth, 7, g_table, 2, 0};
  g_chips[kCpu] = {"cpu", 0, kCpuWidth, kCpuInter, g_cpu, kCpuConstraints, 5};
  g_chips[kCpu2] = {"cpu2", 0, kCpuWidth, kCpuInter, g_cpu, kCpuConstraints, 5};
  {
    static const char* const names[kNumCpuInst] = {"cpu", "cpu2", "cpu3", "cpu4", "cpu5", "cpu6", "cpu7", "cpu8"};
    for (int i = 2; i < kNumCpuInst; ++i) g_chips[cpu_chip(i)] = {names[i], 0, kCpuWidth, kCpuInter, g_cpu, kCpuConstraints, 5};
  }
  g_chips[kEcall] = {"ecall", 0, kEcallWidth, 12, g_ecall, kEcallConstraints, 0};
  g_chips[kKeccak] = {"keccak", 0, kKeccakWidth, 50, g_keccak, kKeccakConstraints, 0};
  g_chips[kKmem] = {"keccak-mem", 0, kKmemWidth, 8, g_kmem, kKmemConstraints, 0};
  g_chips[kMemFinal] = {"mem-final", 0, kMemFinalWidth, 10, g_memfinal, kMemFinalConstraints, 0};
  g_chips[kImage] = {"image", kImagePrepWidth, kImageWidth, 1, g_image, 1, 0};
  g_chips[kProgram] = {"program", kProgramPrepWidth, kProgramWidth, 1, g_program, 0, 0};
  g_chips[kMul] = {"mul", 0, kMulWidth, 5, g_mul, kMulConstraints, 0};
  g_chips[kAlu] = {"alu", 0, kAluWidth, 1, g_alu, kAluConstraints, 0};
  g_chips[kAlu2] = {"alu2", 0, kAluWidth, 1, g_alu, kAluConstraints, 0};
  g_chips[kSub] = {"subword", 0, kSubWidth, 5, g_sub, kSubConstraints, 0};
  g_chips[kSub2] = {"subword2", 0, kSubWidth, 5, g_sub, kSubConstraints, 0};
  g_chips[kBw] = {"bitwise", 0, kBwWidth, 5, g_bw, kBwConstraints, 0};
  g_chips[kBw2] = {"bitwise2", 0, kBwWidth, 5, g_bw, kBwConstraints, 0};
}

}  // namespace

const ChipDef& chip_def(int chip) {
  static std::once_flag once;
  std::call_once(once, build);
  return g_chips[chip];
}

}  // namespace mach
}  // namespace zksp
