// The recorder behind zeta_program.hpp: the AIR templates instantiated over a value type whose arithmetic appends operations to
// a program instead of computing.  Follows verify_machine_proof's "constraint identity at zeta" section step by step (the two
// are cross-checked on real proofs: zksp_zeta_program_selftest).
#include "zeta_program.hpp"

#include <algorithm>
#include <map>
#include <mutex>

#include "context.hpp"
#include "host_hash.hpp"

namespace zksp {
using namespace mach;

namespace {

struct Recorder {
  ZetaProgram* zp = nullptr;
  std::map<uint32_t, uint32_t> consts;  // Montgomery word -> cell
  uint32_t neg1 = 0, two = 0;
  uint32_t fresh() { return zp->n_cells++; }
  uint32_t constant(uint32_t monty) {
    auto it = consts.find(monty);
    if (it != consts.end()) return it->second;
    const uint32_t c = fresh();
    consts[monty] = c;
    zp->const_cell.push_back(c);
    zp->const_monty.push_back(monty);
    return c;
  }
  uint32_t op(uint32_t a, uint32_t b, uint32_t d) {
    const uint32_t c = fresh();
    zp->ops.push_back({a, b, d, c});
    return c;
  }
};
Recorder* g_rec = nullptr;  // (the program is built once, under a lock)

// a value of the recorded computation: the cell that will hold it
struct Sym {
  uint32_t cell;
  Sym operator+(const Sym& o) const { return {g_rec->op(cell, g_rec->zp->one, o.cell)}; }  // a * 1 + o
  Sym operator-(const Sym& o) const { return {g_rec->op(o.cell, g_rec->neg1, cell)}; }     // o * (-1) + a
  Sym operator*(const Sym& o) const { return {g_rec->op(cell, o.cell, g_rec->zp->zero)}; }
  Sym dbl() const { return {g_rec->op(cell, g_rec->two, g_rec->zp->zero)}; }
};
Sym mul_add(Sym a, Sym b, Sym d) { return {g_rec->op(a.cell, b.cell, d.cell)}; }

// what ZetaCtx is to the verifier (mverifier.cpp): the templates' view of one chip
struct RecCtx {
  using F = Sym;
  const ZetaChipCells* cc;
  const std::vector<uint32_t>* apow;  // cells of alpha^(offset + k)
  int k_ = 0;
  Sym acc;
  F prep(int col) const { return {cc->prep + (uint32_t)col}; }
  const P2Consts* p2() const { return &host_p2_consts(); }
  F local(int col) const { return {cc->main + (uint32_t)col}; }
  F next(int col) const { return {cc->main_next + (uint32_t)col}; }
  F is_first() const { return {cc->first}; }
  F is_trans() const { return {cc->trans}; }
  F is_last() const { return {cc->last}; }
  F pub(int which) const { return {cc->pub[which]}; }
  F one() const { return {g_rec->zp->one}; }
  F k(uint32_t monty) const { return {g_rec->constant(monty)}; }
  void emit(F v) { acc = mul_add({(*apow)[(size_t)k_++]}, v, acc); }
  void emit_at(int idx, F v) { acc = mul_add({(*apow)[(size_t)idx]}, v, acc); }
  void set_count(int n) { k_ = n; }
  Sym stash_[32];
  void stash(int i, F v) { stash_[i] = v; }
  F stashed(int i) const { return stash_[i]; }
};

// four opened base columns = one extension column: v0 + v1 X + v2 X^2 + v3 X^3
Sym from_basis(uint32_t first_cell) {
  const ZetaProgram& zp = *g_rec->zp;
  Sym r{first_cell};
  for (uint32_t j = 1; j < 4; ++j) r = mul_add({first_cell + j}, {zp.basis[j]}, r);
  return r;
}

// A product that only feeds one addition becomes that addition's own product: x * y (+ 0), then t * 1 + d (or e * 1 + t), is
// x * y + d - the chip's row is a multiply-add, and the recorder emits one operation per operator of the templates.
void fuse(ZetaProgram* zp) {
  std::vector<uint32_t> uses(zp->n_cells, 0), producer(zp->n_cells, 0xffffffffu);
  for (size_t i = 0; i < zp->ops.size(); ++i) {
    const ZetaOp& o = zp->ops[i];
    ++uses[o.a]; ++uses[o.b]; ++uses[o.d];
    producer[o.c] = (uint32_t)i;
  }
  for (int c = 0; c < kNumChips; ++c) ++uses[zp->chip[c].acc];
  ++uses[zp->result];
  std::vector<uint8_t> dead(zp->ops.size(), 0);
  auto pure_product = [&](uint32_t cell) -> int64_t {  // the operation that made `cell`, if it is x * y + 0 used once
    const uint32_t i = producer[cell];
    if (i == 0xffffffffu || dead[i] || uses[cell] != 1) return -1;
    const ZetaOp& o = zp->ops[i];
    return o.d == zp->zero && o.b != zp->one && o.a != zp->one ? (int64_t)i : -1;
  };
  std::vector<size_t> removed_of_chip(kNumChips, 0);  // (the per-chip counts follow the operations that remain)
  for (size_t j = 0; j < zp->ops.size(); ++j) {
    ZetaOp& o = zp->ops[j];
    if (o.b != zp->one) continue;  // not an addition
    int64_t i = pure_product(o.a);
    uint32_t other = o.d;
    if (i < 0) { i = pure_product(o.d); other = o.a; }
    if (i < 0) continue;
    const ZetaOp& m = zp->ops[(size_t)i];
    o = {m.a, m.b, other, o.c};
    dead[(size_t)i] = 1;
    // (which chip the removed product belonged to: the one whose range holds it)
    size_t s2 = 0;
    for (int c = 0; c < kNumChips; ++c) {
      if ((size_t)i >= s2 && (size_t)i < s2 + zp->ops_of_chip[c]) { ++removed_of_chip[(size_t)c]; break; }
      s2 += zp->ops_of_chip[c];
    }
  }
  std::vector<ZetaOp> kept;
  kept.reserve(zp->ops.size());
  for (size_t i = 0; i < zp->ops.size(); ++i)
    if (!dead[i]) kept.push_back(zp->ops[i]);
  zp->ops.swap(kept);
  for (int c = 0; c < kNumChips; ++c) zp->ops_of_chip[c] -= removed_of_chip[(size_t)c];
  // read multiplicities of the fused program
  zp->reads.assign(zp->n_cells, 0);
  for (const ZetaOp& o : zp->ops) { ++zp->reads[o.a]; ++zp->reads[o.b]; ++zp->reads[o.d]; }
  for (int c = 0; c < kNumChips; ++c) ++zp->reads[zp->chip[c].acc];
  ++zp->reads[zp->result];
  zp->max_reads = 0;
  zp->inputs_read = 0;
  for (uint32_t c = 0; c < zp->n_cells; ++c) {
    zp->max_reads = std::max(zp->max_reads, zp->reads[c]);
    if (c < zp->n_inputs && zp->reads[c]) ++zp->inputs_read;
  }
}

void build(ZetaProgram* zp) {
  Recorder rec;
  rec.zp = zp;
  g_rec = &rec;
  // ---- input cells first: the challenges, then per chip the opened values and the verifier's constants ----
  zp->alpha = rec.fresh();
  zp->gamma = rec.fresh();
  zp->beta = rec.fresh();
  auto run = [&](int n) { const uint32_t f = zp->n_cells; zp->n_cells += (uint32_t)n; return f; };
  for (int c = 0; c < kNumChips; ++c) {
    const ChipDef& d = chip_def(c);
    ZetaChipCells& cc = zp->chip[c];
    cc.prep = run(d.prep_w);
    cc.main = run(d.main_w);
    cc.main_next = run(d.main_w);
    cc.perm = run(d.perm_width());
    cc.perm_next_phi = run(4);
    cc.quot = run(8);
    cc.first = rec.fresh(); cc.trans = rec.fresh(); cc.last = rec.fresh(); cc.apow0 = rec.fresh(); cc.cum_step = rec.fresh();
    for (int i = 0; i < kNumCpuPub; ++i) cc.pub[i] = rec.fresh();
    cc.kappa = rec.fresh(); cc.u = rec.fresh(); cc.v = rec.fresh();
  }
  zp->n_inputs = zp->n_cells;
  // ---- constants ----
  zp->zero = rec.constant(0);
  zp->one = rec.constant(Fp::one().v);
  rec.neg1 = rec.constant((-Fp::one()).v);
  rec.two = rec.constant(Fp::from_canonical(2).v);
  zp->basis[0] = zp->one;
  for (int j = 1; j < 4; ++j) zp->basis[j] = rec.fresh();  // (extension constants: filled by the interpreter)
  // powers of beta, shared by every fingerprint
  uint32_t bpow[kInterMaxElems + 1];
  bpow[0] = zp->one;
  for (int j = 1; j <= kInterMaxElems; ++j) bpow[j] = rec.op(bpow[j - 1], zp->beta, zp->zero);

  for (int c = 0; c < kNumChips; ++c) {
    const size_t ops_before = zp->ops.size();
    const ChipDef& d = chip_def(c);
    ZetaChipCells& cc = zp->chip[c];
    const int pw = d.prep_w, nh = d.helpers(), nb = d.n_constraints;
    std::vector<uint32_t> apow((size_t)d.total_constraints());
    apow[0] = cc.apow0;
    for (size_t k = 1; k < apow.size(); ++k) apow[k] = rec.op(apow[k - 1], zp->alpha, zp->zero);
    RecCtx zc;
    zc.cc = &cc;
    zc.apow = &apow;
    zc.acc = {zp->zero};
    switch (c) {
      case kCpu:
      case kCpu2: case kCpu3: case kCpu4: case kCpu5: case kCpu6: case kCpu7: case kCpu8: eval_cpu(zc); break;
      case kKeccak:
        for (int task = 0; task < ka::kBusTask; ++task) ka::eval_task(task, zc);
        zc.k_ = ka::kNumConstraints;
        eval_keccak_ts(zc);
        break;
      case kKmem: eval_kmem(zc); break;
      case kMemFinal: eval_memfinal(zc); break;
      case kImage: eval_image(zc); break;
      case kProgram: break;
      case kMul: eval_mul(zc); break;
      case kTable: eval_table(zc); break;
      case kAlu:
      case kAlu2: eval_alu(zc); break;
      case kSub:
      case kSub2: eval_sub(zc); break;
      case kBw:
      case kBw2: eval_bw(zc); break;
      case kP2: eval_p2(zc); break;
      case kEcall: eval_ecall(zc); break;
      case kQr: eval_qr(zc); break;
      case kTr: eval_tr(zc); break;
      case kDiv: eval_div(zc); break;
      case kHint: eval_hint(zc); break;
    }
    // LogUp (machine_defs.hpp "LogUp layout"): row = [prep | main] at zeta
    auto row = [&](int col) -> Sym { return col < pw ? Sym{cc.prep + (uint32_t)col} : Sym{cc.main + (uint32_t)(col - pw)}; };
    auto lf_eval = [&](const LinForm& f) {
      Sym v{rec.constant(f.c0)};
      for (int i = 0; i < f.n; ++i) v = mul_add(row(f.col[i]), {rec.constant(f.coef[i])}, v);
      return v;
    };
    auto fingerprint = [&](const Interaction& it) {
      Sym f = mul_add({zp->gamma}, {zp->one}, {rec.constant(Fp::from_canonical((uint32_t)it.bus).v)});
      for (int j = 0; j < it.n_el; ++j) f = mul_add({bpow[j + 1]}, lf_eval(it.el[j]), f);
      return f;
    };
    auto signed_mult = [&](const Interaction& it) {
      const Sym m = lf_eval(it.mult);
      return it.sign < 0 ? mul_add(m, {rec.neg1}, {zp->zero}) : m;
    };
    const int nr = d.n_inter - d.n_merged, npairs = (nr + 1) / 2;
    // v fa fb - (ma fb + mb fa) for slot s and a candidate value v
    auto slot_constraint = [&](int s, Sym v) {
      Sym ma{zp->zero}, fa{zp->zero}, mb{zp->zero}, fb{zp->one};
      if (s < npairs) {
        ma = signed_mult(d.inter[2 * s]);
        fa = fingerprint(d.inter[2 * s]);
        if (2 * s + 1 < nr) {
          mb = signed_mult(d.inter[2 * s + 1]);
          fb = fingerprint(d.inter[2 * s + 1]);
        }
      } else {  // the merged sends: (sum m_k) / (sum m_k f_k + 1 - sum m_k)
        for (int k = nr; k < d.n_inter; ++k) {
          const Sym m = signed_mult(d.inter[k]);
          ma = ma + m;
          fa = mul_add(m, fingerprint(d.inter[k]), fa);
        }
        fa = fa + (Sym{zp->one} - ma);
      }
      return v * fa * fb - (mul_add(fb, ma, fa * mb));
    };
    Sym hsum{zp->zero};
    for (int j = 0; j < nh; ++j) {
      const Sym hj = from_basis(cc.perm + 4 * (uint32_t)j);
      hsum = hsum + hj;
      zc.acc = mul_add({apow[(size_t)(nb + j)]}, slot_constraint(j, hj), zc.acc);
    }
    const Sym phi = from_basis(cc.perm + 4 * (uint32_t)nh), phin = from_basis(cc.perm_next_phi);
    zc.acc = mul_add({apow[(size_t)(nb + nh)]}, slot_constraint(nh, phin - phi + Sym{cc.cum_step} - hsum), zc.acc);
    cc.acc = zc.acc.cell;
    zp->ops_of_chip[c] = zp->ops.size() - ops_before;
  }
  // ---- the final combination: one random linear combination over the heights' quotients ----
  Sym r{zp->zero};
  for (int c = 0; c < kNumChips; ++c) {
    const ZetaChipCells& cc = zp->chip[c];
    r = mul_add({cc.kappa}, {cc.acc}, r);
    const Sym q0 = from_basis(cc.quot), q1 = from_basis(cc.quot + 4);
    r = r - Sym{cc.u} * q0;
    r = mul_add({cc.v}, q1, r);
  }
  zp->result = r.cell;
  g_rec = nullptr;
  fuse(zp);
}

}  // namespace

const ZetaProgram& zeta_program() {
  static ZetaProgram zp;
  static std::once_flag once;
  std::call_once(once, [] { build(&zp); });
  return zp;
}

void zeta_program_run(const ZetaProgram& zp, Fp4* cells) {
  for (size_t i = 0; i < zp.const_cell.size(); ++i) cells[zp.const_cell[i]] = Fp4::from_base(Fp::raw(zp.const_monty[i]));
  for (int j = 1; j < 4; ++j) {
    Fp4 b = Fp4::zero();
    b.c[j] = Fp::one();
    cells[zp.basis[j]] = b;
  }
  for (const ZetaOp& o : zp.ops) cells[o.c] = cells[o.a] * cells[o.b] + cells[o.d];
}

bool zeta_program_memory_balances(const ZetaProgram& zp, const Fp4* cells, const Fp4& gamma, const Fp4& beta) {
  Fp4 bp[5];
  bp[0] = beta;
  for (int j = 1; j < 5; ++j) bp[j] = bp[j - 1] * beta;
  auto fp = [&](uint32_t cell) {
    Fp4 f = gamma + bp[0] * Fp::from_canonical(cell);
    for (int j = 0; j < 4; ++j) f += bp[j + 1] * cells[cell].c[j];
    return f;
  };
  // (one inversion per cell: reads of a cell share its fingerprint)
  Fp4 total = Fp4::zero();
  std::vector<uint8_t> written(zp.n_cells, 0);
  for (uint32_t c = 0; c < zp.n_inputs; ++c) written[c] = 1;
  for (uint32_t c : zp.const_cell) written[c] = 1;
  for (int j = 1; j < 4; ++j) written[zp.basis[j]] = 1;
  std::vector<uint32_t> seen(zp.n_cells, 0);
  for (const ZetaOp& o : zp.ops) {
    if (!written[o.a] || !written[o.b] || !written[o.d] || written[o.c]) return false;  // read before written, or written twice
    written[o.c] = 1;
    ++seen[o.a]; ++seen[o.b]; ++seen[o.d];
  }
  for (int c = 0; c < kNumChips; ++c) ++seen[zp.chip[c].acc];
  ++seen[zp.result];
  for (uint32_t c = 0; c < zp.n_cells; ++c) {
    if (!zp.reads[c] && !seen[c]) continue;
    const Fp4 inv = fp(c).inv();
    total += inv * Fp::from_canonical(zp.reads[c]);  // the write, with the program's multiplicity
    total -= inv * Fp::from_canonical(seen[c]);      // the reads that happened
  }
  return total == Fp4::zero();
}

}  // namespace zksp
