// zksp_prove / zksp_prove_batch: replaces `client.prove(&pk, stdin).run()`
// (reference prover/src/bin/main.rs:71-74) for one input or many.
//
// The batch path is a three-stage pipeline over "waves" of inputs:
//   host threads   run the guest (RV32IM executor) for wave k+1
//   the GPU        proves wave k (one fixed launch sequence, bodies -> pinned staging)
//   host threads   build the proof objects of wave k-1
// so that neither the executor nor the proof assembly sits on the critical path
// once a few waves are in flight.
#include "../../../include/zksp.h"
#include "../../../include/zksp_component.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <new>
#include <thread>

#include "api_types.hpp"
#include "machine_defs.hpp"
#include "mprover.hpp"
#ifdef ZKSP_COMPONENT
#include "prover.hpp"
#endif

using namespace zksp;

namespace {

int ceil_log2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}

#ifdef ZKSP_COMPONENT  // the round-1 keccak-chip component path (include/zksp_component.h): not in the default library
int trace_log_height(size_t n_perms) { return ceil_log2(std::max<size_t>(24 * n_perms, 32)); }

struct Job {
  ExecutionRecord rec;
  int logh = 0;
};

// one launch group: proofs of one trace height proven in lockstep
struct Group {
  int logh = 0;
  std::vector<size_t> idxs;   // input indices
  std::vector<uint32_t> np;   // permutations per proof
  size_t bw = 0;              // body words per proof
  int slot = 0;               // staging buffer holding its bodies
};
#endif

// Host threads this process may use for guest execution and proof assembly.  A GPU box shows
// every core of its host (hardware_concurrency() = 256) while one process per GPU owns a share of
// them, so the default is bounded; ZKSP_HOST_THREADS overrides it.
unsigned host_threads() {
  static const unsigned n = [] {
    if (const char* e = getenv("ZKSP_HOST_THREADS")) {
      const int v = atoi(e);
      if (v > 0) return (unsigned)v;
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return std::min(hw, 16u);
  }();
  return n;
}

// ZKSP_TRACE_BATCH=1: wall-clock marks of the batch pipeline on stderr (debugging aid)
struct BatchTrace {
  bool on = getenv("ZKSP_TRACE_BATCH") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what, size_t k) const {
    if (!on) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[zksp batch] %8.2f ms  %s %zu\n", ms, what, k);
  }
};

void parallel_for(size_t count, unsigned max_threads, const std::function<void(size_t)>& fn) {
  std::atomic<size_t> next{0};
  unsigned nt = std::max(1u, std::min<unsigned>(std::min(max_threads, host_threads()), (unsigned)count));
  auto work = [&]() {
    for (size_t j; (j = next.fetch_add(1)) < count;) fn(j);
  };
  std::vector<std::thread> th;
  try {
    th.reserve(nt);
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
  } catch (...) {
  }  // fewer threads than asked for: the items are claimed from one counter, so the ones that exist finish the job
  work();
  for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

// Machine proofs (the full statement): trace every guest run on the host threads, group runs of similar size
// (one shape - the chip heights of the group's largest counts - per group), prove each group in lockstep on the GPU,
// wrap the bodies into proof objects.
static int prove_batch_machine(zksp_client* c, const zksp_pk* pk, zksp_stdin* const* stdins, size_t n, zksp_proof** out,
                               int32_t* status) {
  Context* ctx = &c->ctx;
  for (size_t i = 0; i < n; ++i) { out[i] = nullptr; status[i] = ZKSP_ERR_INVALID_ARG; }
  if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
    return ctx->fail(ZKSP_ERR_HIP, "prove: could not create the copy stream");
  for (int k = 0; k < 2; ++k)
    if ((!ctx->ev_proved[k] && hipEventCreateWithFlags(&ctx->ev_proved[k], hipEventDisableTiming) != hipSuccess) ||
        (!ctx->ev_copied[k] && hipEventCreateWithFlags(&ctx->ev_copied[k], hipEventDisableTiming) != hipSuccess))
      return ctx->fail(ZKSP_ERR_HIP, "prove: could not create the copy events");
  ctx->body_free = nullptr;
  // a call of several chunks copies records and bodies beside its passes: the side lanes then carry small batches only
  // (Context::lane_max_batch); a call that fits one chunk proves like a resident pass
  struct PipelinedScope {
    Context* c;
    ~PipelinedScope() { c->pipelined = false; }
  } pipelined_scope{ctx};
  ctx->pipelined = n > (size_t)Context::kSideMaxBatch;
  std::vector<std::unique_ptr<zksp_mtrace>> traces(n);
  const BatchTrace mark;
  // threads a run's deferred leaf checks may use: the host's threads over the runs traced at once (the first wave is few
  // runs whose latency the GPU waits for; the rest are many, traced for throughput)
  std::atomic<unsigned> leaf_budget{2};
  auto trace_one = [&](size_t i) {
    if (!stdins[i]) return;
    try {
      traces[i].reset(new zksp_mtrace());
      traces[i]->prog = &pk->mprog;
      if (!stdins[i]->deferred.empty()) {
        // the run's leaf checks, deferred to this call: made here, on a tracing thread, while the GPU proves the runs before
        traces[i]->leaf_rc = stdin_resolve_deferred(c, stdins[i], &traces[i]->leaf_err, leaf_budget.load(std::memory_order_relaxed));
        if (traces[i]->leaf_rc) return;
      }
      if (stdins[i]->leaf_check) stdins[i]->statement = stdins[i]->leaf_check->pub_tuples;  // (kept: proving consumes the checks)
      trace_execute(pk->elf, pk->mprog, stdins[i]->entries, (uint64_t)1 << 21, &traces[i]->t);
      traces[i]->t.agg_leaves.swap(stdins[i]->agg_leaves);
      traces[i]->t.agg_keys.swap(stdins[i]->agg_keys);
      traces[i]->t.agg_rows = machine_agg_row_count(traces[i]->t.agg_keys.empty() ? nullptr : traces[i]->t.agg_keys.data(),
                                                    traces[i]->t.agg_leaves.size() / 8);
      traces[i]->t.leaf_check = std::move(stdins[i]->leaf_check);
      stdins[i]->leaf_check.reset();
      stdins[i]->entries.clear();  // consumed, as SP1Stdin is by prove()
    } catch (...) {
      traces[i].reset(new zksp_mtrace());
      traces[i]->t.rec.error = "out of memory while tracing the guest";
    }
  };
  // A first wave (an eighth of the call, at most max_batch) is traced before anything else so that the GPU starts early;
  // the rest are traced by host threads while it proves them.
  // (an eighth, at most sixty-four: sixty-four proofs already run at 95 % of the rate of a full batch, and they are traced
  // and uploaded in a tenth of a second)
  // (round 5: a large call starts with 24 and lets the waves grow - the GPU starts after 40 ms instead of 135, and the next
  // wave is uploaded, from pageable memory at about 10 GB/s, in about the time the wave before it is proven;
  // measured on 1 024 acct-d8 runs the warm call is 3 034 ms with the ramp and 3 042 ms without: kept, it costs nothing)
  // (runs that bring deferred leaf checks - the nodes of a recursion tree - are traced at the pace of those checks, some 15 ms
  // of all the host's cores each: the first wave is four of them, the later ones eight, so that the GPU starts early and never
  // waits for a long wave's checks)
  bool any_deferred = false;
  for (size_t i = 0; i < n; ++i) any_deferred |= stdins[i] && !stdins[i]->deferred.empty();
  const bool ramp = n >= 384 && ctx->params.max_batch >= 96 && !any_deferred;
  const size_t w0 = any_deferred ? std::min<size_t>(n, 4)
                    : ramp ? 24 : std::min<size_t>(n, std::max<size_t>(1, std::min<size_t>(std::min<size_t>(ctx->params.max_batch, 64), std::max<size_t>(16, (n + 7) / 8))));
  leaf_budget = std::max(2u, host_threads() / (unsigned)std::max<size_t>(1, std::min<size_t>(w0, host_threads())));
  parallel_for(w0, 64, trace_one);
  leaf_budget = 2;
  mark.mark("traced", w0);
  std::string first_err;
  int rc_all = ZKSP_OK;

  // A chunk: at most `cap` runs proven in lockstep with one shape.  Runs are grouped by size class (the heights of the
  // chips that are not split, and the binary order of magnitude of the cycle / ALU / sub-word counts); a group's shape
  // covers its largest counts, so runs whose counts straddle a power of two still share a batch.
  struct Chunk {
    std::array<int, mach::kNumChips> lh;
    std::vector<size_t> idx;
    size_t group_size;  // runs of these heights in the same window (sizes the workspace once)
  };
  std::vector<Chunk> chunks;
  size_t last_cap = 0;   // the largest chunk the HBM budget allowed the last group that was cut into chunks
  size_t size_hint = 0;  // the largest chunk a homogeneous call will load: the workspace is sized for it at the first load
  auto build_chunks = [&](size_t lo, size_t hi) {
    std::map<std::array<int, mach::kNumChips>, std::vector<size_t>> groups;
    std::map<std::array<int, mach::kNumChips>, MachineCounts> covers;
    auto clog2 = [](size_t v) { int l = 0; while (((size_t)1 << l) < v) ++l; return l; };
    for (size_t i = lo; i < hi; ++i) {
      if (!stdins[i] || !traces[i]) continue;
      if (traces[i]->leaf_rc) {
        status[i] = traces[i]->leaf_rc;
        if (first_err.empty()) first_err = traces[i]->leaf_err;
        continue;
      }
      const ExecutionRecord& r = traces[i]->t.rec;
      if (!r.error.empty() || !r.halted) {
        status[i] = ZKSP_ERR_EXECUTOR;
        if (first_err.empty()) first_err = "executor: " + (r.error.empty() ? std::string("guest did not halt") : r.error);
        continue;
      }
      if (r.exit_code != 0) {
        status[i] = ZKSP_ERR_GUEST_PANIC;
        if (first_err.empty()) first_err = "guest panicked (exit code " + std::to_string(r.exit_code) + "): " + r.stderr_text;
        continue;
      }
      std::array<int, mach::kNumChips> lh;
      machine_heights(pk->mprog, traces[i]->t, lh.data());
      const MachineTrace& t = traces[i]->t;
      for (int k = 1; k < mach::kNumCpuInst; ++k) lh[mach::cpu_chip(k)] = 0;  // (the key: the instances' common height)
      lh[mach::kAlu] = clog2(t.alu_idx.size()); lh[mach::kAlu2] = 0;
      lh[mach::kSub] = clog2(t.sub_idx.size()); lh[mach::kSub2] = 0;
      lh[mach::kBw] = clog2(t.bw_idx.size()); lh[mach::kBw2] = 0;
      lh[mach::kP2] = clog2(t.p2_rows() + 1);
      lh[mach::kQr] = clog2(t.qr_rows() + 1);
      lh[mach::kTr] = clog2(t.tr_rows() + 1);
      lh[mach::kHint] = 0;
      // (the small chips do not split a group: a run with 513 multiplications shares the shape of one with 511, the
      // cover gives both the taller multiplier chip)
      lh[mach::kEcall] = 0;
      lh[mach::kMul] = 0;
      lh[mach::kDiv] = 0;
      groups[lh].push_back(i);
      covers[lh].cover(t);
    }
    // A group of a few runs costs a workspace layout, an upload the GPU waits for and a pass at a poor rate: it joins the
    // group it disturbs least, if the padding that costs (both groups are then proven with the cover of their counts)
    // stays below the work of eight proofs.  A block of 300 receipts goes from eleven shapes to five.
    auto cells_of = [&](const MachineCounts& cnt) {
      int hh[mach::kNumChips];
      machine_heights(pk->mprog, cnt, hh);
      size_t cells = 0;
      for (int ch = 0; ch < mach::kNumChips; ++ch) {
        const mach::ChipDef& d = mach::chip_def(ch);
        cells += (size_t)(d.main_w + d.perm_width() + 8) << hh[ch];
      }
      return cells;
    };
    for (bool merged = true; merged && groups.size() > 1;) {
      merged = false;
      for (auto it = groups.begin(); it != groups.end(); ++it) {
        if (it->second.size() >= 16) continue;
        const MachineCounts cg = covers[it->first];
        const size_t cells_g = cells_of(cg);
        size_t best_cost = SIZE_MAX;
        auto best = groups.end();
        for (auto jt = groups.begin(); jt != groups.end(); ++jt) {
          if (jt == it) continue;
          MachineCounts m = covers[jt->first];
          const size_t cells_h = cells_of(m);
          m.cover(cg);
          const size_t cm = cells_of(m), cost = it->second.size() * (cm - cells_g) + jt->second.size() * (cm - cells_h);
          if (cost < best_cost && cost <= 8 * cm) { best_cost = cost; best = jt; }
        }
        if (best == groups.end()) continue;
        best->second.insert(best->second.end(), it->second.begin(), it->second.end());
        covers[best->first].cover(cg);
        covers.erase(it->first);
        groups.erase(it);
        merged = true;
        break;  // (the iterator is gone: start over)
      }
    }
    for (auto& kv0 : groups) {
      std::pair<std::array<int, mach::kNumChips>, std::vector<size_t>> kv;
      machine_heights(pk->mprog, covers[kv0.first], kv.first.data());
      kv.second = kv0.second;
      // bytes of HBM one proof of these heights needs (traces, coefficients, LDEs of the three rounds, scratch)
      size_t per_proof = 0;
      for (int ch = 0; ch < mach::kNumChips; ++ch) {
        const mach::ChipDef& d = mach::chip_def(ch);
        per_proof += ((size_t)(d.main_w + d.perm_width() + 8) * 16 + 64) << kv.first[ch];
      }
      // 16 bytes per cell already cover trees, FRI layers and scratch (measured: an acct-d8 proof takes 702 MiB of arena where
      // this sum says 900); a seventh more for safety.  The budget is what the device has free right now plus what this
      // client's arena already holds, less a fifth for everybody else
      per_proof += per_proof / 7;
      size_t mem_free = 0, mem_total = 0;
      if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = (size_t)64 << 30;
      const size_t budget = (mem_free + ctx->arena_bytes) / 5 * 4;
      // ... and at least two chunks behind the first wave where the call is large enough, so that the upload of one chunk
      // overlaps the proving of another (the bodies are fetched behind the next pass and wrapped on several threads, so
      // more, smaller chunks only lower the rate of a pass)
      const size_t pipe = std::max<size_t>(16, (n + 1) / 2);
      const size_t cap = std::max<size_t>(1, std::min({(size_t)ctx->params.max_batch, budget / std::max<size_t>(per_proof, 1), pipe}));
      last_cap = cap;
      // chunks of equal size (576 runs under a cap of 174 go as 4 x 144, not 3 x 174 + 54: a small last chunk proves at a
      // poor rate)
      const size_t m = kv.second.size(), nck = (m + cap - 1) / cap, per = (m + nck - 1) / nck;
      for (size_t off = 0; off < m; off += per) {
        Chunk ck;
        ck.lh = kv.first;
        ck.group_size = std::min(per, m);
        ck.idx.assign(kv.second.begin() + off, kv.second.begin() + std::min(off + per, m));
        chunks.push_back(std::move(ck));
      }
    }
  };

  // Traces whose records are on the device keep only the execution record (the proof objects need nothing else):
  // the bulky vectors go to a helper thread that frees them (tens of megabytes each) while the GPU works.
  struct Reaper {
    std::vector<std::thread> th;
    ~Reaper() {
      for (auto& t : th)
        if (t.joinable()) t.join();
    }
  } reaper;
  // (`after`: the event behind an upload nobody has waited for yet, or null)
  auto release_chunk = [&](const Chunk& ck, hipEvent_t after) {
    std::vector<MachineTrace> dead(ck.idx.size());
    for (size_t j = 0; j < ck.idx.size(); ++j) {
      MachineTrace& t = traces[ck.idx[j]]->t;
      dead[j].cycles.swap(t.cycles); dead[j].keccak.swap(t.keccak); dead[j].memfinal.swap(t.memfinal);
      dead[j].muls.swap(t.muls); dead[j].prog_mult.swap(t.prog_mult); dead[j].alu_idx.swap(t.alu_idx); dead[j].sub_idx.swap(t.sub_idx); dead[j].bw_idx.swap(t.bw_idx); dead[j].div_idx.swap(t.div_idx);
    }
    auto bury = [after](std::vector<MachineTrace>& d) {
      if (after) (void)hipEventSynchronize(after);
      d.clear();
    };
    try {
      reaper.th.emplace_back([bury, d = std::move(dead)]() mutable { bury(d); });
    } catch (...) {
      bury(dead);  // no thread: here instead
    }
  };
  // events behind the asynchronous uploads of this call (destroyed when the call is over and the reaper has been joined)
  std::vector<hipEvent_t> load_events;
  auto load_chunk = [&](const Chunk& ck, bool into_spare) {
    std::vector<const MachineTrace*> ts(ck.idx.size());
    for (size_t j = 0; j < ck.idx.size(); ++j) {
      ts[j] = &traces[ck.idx[j]]->t;
      for (int k = 1; k < mach::kNumCpuInst; ++k)
        traces[ck.idx[j]]->handover_pc[k - 1] = machine_handover_pc(pk->mprog, *ts[j], ck.lh.data(), k);
    }
    if (!into_spare) ctx->batch_hint = (int)std::max(ck.group_size, size_hint);
    else ctx->spare_batch_hint = (int)std::max(ck.group_size, size_hint);
    hipEvent_t ev = nullptr;
    if (into_spare) {  // not waited for: the proving stream and the thread that frees the traces wait for this event
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        try {
          load_events.push_back(ev);
        } catch (...) {
          (void)hipEventDestroy(ev);
          ev = nullptr;
        }
      } else {
        ev = nullptr;
      }
    }
    const int rc = machine_load(ctx, pk->mprog, pk->mvk, ts.data(), ts.size(), into_spare, ck.lh.data(), ev);
    mark.mark(into_spare ? "next enqueued" : "loaded", ts.size());
    if (!into_spare && ctx->mws) mark.mark("arena MiB per proof of the workspace", (size_t)(ctx->arena_bytes >> 20) / (size_t)std::max(1, ctx->mws->batch));
    if (rc == ZKSP_OK) release_chunk(ck, ev);
    else if (ev) (void)hipStreamSynchronize(ctx->copy_stream);  // whatever was enqueued reads the traces: wait before they can go
    return rc;
  };
  auto fail_chunk = [&](const Chunk& ck, int rc) {
    for (size_t i : ck.idx) status[i] = rc;
    rc_all = rc;
  };

  // declared after `traces` and `reaper`: joined before either is destroyed, on every way out
  struct Joiner {
    std::thread t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } rest;
  build_chunks(0, w0);
  // The rest is traced by helper threads, in input order, while the GPU proves; it is GROUPED in waves.  A call whose first
  // wave came out with one shape (a homogeneous workload) goes on wave by wave - a second wave of twice the first, then
  // waves of a full chunk - each grouped as soon as its runs are traced, so that the GPU never waits for more than the next
  // chunk's traces (the host threads trace about three times as fast as the GPU proves).  A mixed workload keeps one global
  // grouping of everything behind the first wave (waves grouped one by one would fragment into small chunks).
  std::vector<size_t> wave_end;  // exclusive ends of the waves behind the first
  if (w0 < n) {
    if (chunks.size() == 1 && n >= 4 * w0) {
      const size_t cap = any_deferred ? 8 : std::max<size_t>(16, std::min<size_t>(ctx->params.max_batch, 192));
      size_t e0 = w0, biggest = 2 * w0;
      if (ramp) {
        // (growth by 7/4: 24, 42, 73, 127 runs, then waves of a full chunk.  Measured against 3/2 - 24, 36, 54, 81, 121, 181 -
        // on 1 024 acct-d8 runs, four calls each: 3 047-3 102 ms against 3 099-3 137; a first wave of 32 or 40 changes nothing)
        for (size_t w = w0 * 7 / 4; w < cap && e0 + w + cap <= n; w = w * 7 / 4) {
          e0 += w;
          wave_end.push_back(e0);
          biggest = w;
        }
      } else {
        e0 = 3 * w0;
        wave_end.push_back(e0);
      }
      const size_t left = n - e0;
      const size_t nw = (left + cap - 1) / cap, per = (left + nw - 1) / nw;  // waves of equal size
      for (size_t e = e0 + per; e < n; e += per) wave_end.push_back(e);
      wave_end.push_back(n);
      size_hint = std::min(std::max(per, biggest), std::max<size_t>(last_cap, 1));
    } else {
      wave_end.push_back(n);
    }
  }
  std::unique_ptr<std::atomic<uint8_t>[]> traced(new std::atomic<uint8_t>[n]);
  for (size_t i = 0; i < n; ++i) traced[i].store(i < w0 ? 1 : 0, std::memory_order_relaxed);
  auto trace_rest = [&]() {
    parallel_for(n - w0, 64, [&](size_t j) {  // (the runs are claimed from one counter: traced in input order)
      trace_one(w0 + j);
      traced[w0 + j].store(1, std::memory_order_release);
    });
  };
  if (w0 < n) {
    try {
      rest.t = std::thread(trace_rest);
    } catch (...) {
      trace_rest();  // no thread to spare: trace them here
    }
  }
  size_t next_wave = 0, grouped = w0;
  auto have_chunk = [&](size_t k) {  // does chunk k exist?  (later runs are grouped once their wave's traces are complete)
    while (k >= chunks.size() && next_wave < wave_end.size()) {
      const size_t hi = wave_end[next_wave];
      if (next_wave + 1 == wave_end.size()) {
        if (rest.t.joinable()) rest.t.join();
      } else {
        for (size_t i = grouped; i < hi; ++i)
          while (!traced[i].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
      mark.mark("wave traced", hi - grouped);
      build_chunks(grouped, hi);
      grouped = hi;
      ++next_wave;
    }
    return k < chunks.size();
  };
  // Chunk k is proven while chunk k + 1 (same heights) is uploaded into the spare record set and chunk k - 1 is
  // wrapped into proof objects; a chunk of other heights waits for the GPU and takes the plain path.
  struct Joiner2 {
    std::thread t;
    ~Joiner2() {
      if (t.joinable()) t.join();
    }
  } wrapper;  // (declared after everything its thread reads)
  // (the first two waves of a ramped call - 24 and 36 runs - still take the side lanes: measured on 1 024 acct-d8 runs, 3 100 ms
  // against 3 127-3 147 without, and 3 167-3 195 with the third wave - 54 runs - on the lanes as well)
  const size_t ramp_lanes = ramp ? 2 : 0;
  bool in_flight = false;  // chunk k's records are resident and its proving pass is enqueued
  for (size_t k = 0; have_chunk(k); ++k) {
    const Chunk ck = chunks[k];  // a copy: grouping the rest may reallocate `chunks`
    if (!in_flight) {
      int rc = load_chunk(ck, false);
      ctx->pipelined = n > (size_t)Context::kSideMaxBatch && !(k < ramp_lanes);
      if (rc == ZKSP_OK) rc = machine_prove_resident(ctx);
      if (rc != ZKSP_OK) { fail_chunk(ck, rc); continue; }
    }
    in_flight = false;
    const bool more = have_chunk(k + 1);
    // (a chunk of ANOTHER shape is uploaded behind the current pass too - its records live outside the arena - and the
    // arena is laid out for it when it is activated: a block of receipts goes through five shapes without the GPU waiting
    // for an upload)
    const bool piggyback = more && ctx->mws && ctx->copy_stream;
    int rc_next = ZKSP_OK;
    if (piggyback) rc_next = load_chunk(chunks[k + 1], true);
    // The bodies of this pass go to pinned host memory on the copy stream, behind the pass; the NEXT pass is enqueued at
    // once and waits for the copy only before its assemble kernel (Context::body_free), so the GPU does not idle while
    // half a gigabyte of proof bodies crosses PCIe.
    const size_t cnt = ck.idx.size(), bw = ctx->mws->body_words;
    const int slot = (int)(k & 1);
    bool copy_ok = true;
    if (ctx->h_stage2_words[slot] < cnt * bw) {
      if (ctx->h_stage2[slot]) (void)hipHostFree(ctx->h_stage2[slot]);
      ctx->h_stage2[slot] = nullptr;
      ctx->h_stage2_words[slot] = 0;
      if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage2[slot]), cnt * bw * 4, hipHostMallocDefault) == hipSuccess)
        ctx->h_stage2_words[slot] = cnt * bw;
      else
        (void)hipGetLastError();  // no pinned memory to be had: a shortage of the host, not a lost device
    }
    // the staging buffer: pinned if there is one of that size, else pageable memory (the copy is then not overlapped)
    uint32_t* stage = ctx->h_stage2_words[slot] >= cnt * bw ? ctx->h_stage2[slot] : nullptr;
    if (!stage) {
      try {
        ctx->h_stage2_pageable[slot].resize(cnt * bw);
        stage = ctx->h_stage2_pageable[slot].data();
      } catch (...) {
        copy_ok = false;
      }
    }
    copy_ok = copy_ok && hipEventRecord(ctx->ev_proved[slot], ctx->stream) == hipSuccess &&
              hipStreamWaitEvent(ctx->copy_stream, ctx->ev_proved[slot], 0) == hipSuccess &&
              hipMemcpyAsync(stage, ctx->mws->body, cnt * bw * 4, hipMemcpyDeviceToHost, ctx->copy_stream) == hipSuccess &&
              hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream) == hipSuccess;
    if (copy_ok) ctx->body_free = ctx->ev_copied[slot];
    if (copy_ok && piggyback) {  // the GPU goes on with the next chunk while this one is fetched and wrapped
      if (rc_next == ZKSP_OK) rc_next = machine_activate_spare(ctx);
      ctx->pipelined = n > (size_t)Context::kSideMaxBatch && !(k + 1 < ramp_lanes);
      if (rc_next == ZKSP_OK) rc_next = machine_prove_resident(ctx);
      if (rc_next == ZKSP_OK) in_flight = true;
    }
    if (!copy_ok || hipEventSynchronize(ctx->ev_copied[slot]) != hipSuccess) {
      // the device is gone: nothing after this chunk can be proven either (the next chunk's traces may already
      // have been released to the spare upload)
      ctx->body_free = nullptr;
      const int rc = ctx->fail(ZKSP_ERR_HIP, "prove: device-to-host copy failed");
      (void)have_chunk(chunks.size());  // group whatever was traced later, so that it gets a status too
      for (size_t j = k; j < chunks.size(); ++j) fail_chunk(chunks[j], rc);
      break;
    }
    mark.mark("proved and fetched", cnt);
    const uint32_t* bodies = stage;
    // A proof object is 2.6 MB of copying and header work.  The chunk is wrapped on helper threads while this thread goes
    // on to upload the next one (the upload of a chunk and the wrapping of another, one after the other, took longer than
    // the GPU needs for a pass).  One wrap is outstanding at a time: the staging slot it reads is written again two chunks
    // later, after the wrap of the chunk in between has been started - i.e. after this one has been joined.
    if (wrapper.t.joinable()) wrapper.t.join();
    auto wrap = [&, ck, bodies, bw, cnt]() {
      parallel_for(cnt, 8, [&](size_t j) {
        const size_t i = ck.idx[j];
        status[i] = machine_proof_from_parts(pk, traces[i]->t.rec, ck.lh.data(), traces[i]->handover_pc, traces[i]->t.agg_leaves, traces[i]->t.agg_keys,
                                             traces[i]->t.leaf_check.get(), bodies + j * bw, bw, &out[i]);
      });
      mark.mark("wrapped", cnt);
    };
    try {
      wrapper.t = std::thread(wrap);
    } catch (...) {
      wrap();
    }
    if (piggyback && rc_next != ZKSP_OK) {
      fail_chunk(chunks[k + 1], rc_next);
      ++k;  // that chunk is lost (its traces may already be released): go on with the one after it
    }
  }
  if (wrapper.t.joinable()) wrapper.t.join();
  if (!first_err.empty() && rc_all == ZKSP_OK) ctx->error = first_err;
  ctx->body_free = nullptr;  // every copy has completed: later passes on this client need no wait
  ctx->retired.clear();      // ... and every pass has: the workspaces replaced on the way can go
  for (auto& v : ctx->h_stage2_pageable) std::vector<uint32_t>().swap(v);
  mark.mark("done", n);
  for (auto& t : reaper.th)
    if (t.joinable()) t.join();
  mark.mark("reaper joined", reaper.th.size());
  for (hipEvent_t ev : load_events) (void)hipEventDestroy(ev);
  // what is left of the traces (execution records: a few hundred kilobytes each) is freed by a helper thread the next
  // call, or the client's destruction, joins: a tenth of a second per 512 runs that the caller does not wait for
  if (ctx->cleanup.joinable()) ctx->cleanup.join();
  auto* dead = new std::vector<std::unique_ptr<zksp_mtrace>>(std::move(traces));
  try {
    ctx->cleanup = std::thread([dead]() { delete dead; });
  } catch (...) {
    delete dead;
  }
  mark.mark("traces handed over", n);
  return rc_all;
}

static int prove_batch_impl(zksp_client* c, const zksp_pk* pk, zksp_stdin* const* stdins, size_t n, zksp_proof** out,
                            int32_t* status) {
  if (!c || !pk || !stdins || !out || !status || n == 0) return ZKSP_ERR_INVALID_ARG;
  if (c->ctx.params.proof_mode == ZKSP_PROOF_MACHINE) {
    if (!c->ctx.has_device())
      return c->ctx.fail(ZKSP_ERR_NO_DEVICE, "prove: this client was created without a GPU; there is no CPU proving path");
    if (hipSetDevice(c->ctx.device) != hipSuccess) return c->ctx.fail(ZKSP_ERR_HIP, "prove: hipSetDevice failed");
    return prove_batch_machine(c, pk, stdins, n, out, status);
  }
#ifndef ZKSP_COMPONENT
  return c->ctx.fail(ZKSP_ERR_UNSUPPORTED, "prove: this library was built without the keccak-chip component path");
#else
  Context* ctx = &c->ctx;
  if (!ctx->has_device())
    return ctx->fail(ZKSP_ERR_NO_DEVICE, "prove: this client was created without a GPU; there is no CPU proving path");
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(ZKSP_ERR_HIP, "prove: hipSetDevice failed");
  for (size_t i = 0; i < n; ++i) { out[i] = nullptr; status[i] = ZKSP_ERR_INVALID_ARG; }

  // copy stream and events (created on first use, owned by the context); before any thread
  // is started, so that no early return can leave a joinable std::thread behind
  bool overlap = true;
  if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) overlap = false;
  for (int k = 0; k < 2 && overlap; ++k) {
    if (!ctx->ev_proved[k] && hipEventCreateWithFlags(&ctx->ev_proved[k], hipEventDisableTiming) != hipSuccess) overlap = false;
    if (!ctx->ev_copied[k] && hipEventCreateWithFlags(&ctx->ev_copied[k], hipEventDisableTiming) != hipSuccess) overlap = false;
  }
  if (!overlap) return ctx->fail(ZKSP_ERR_HIP, "prove: could not create the copy stream");

  // ---- stage 1: executor workers (run ahead of the GPU, in input order) ----
  std::vector<Job> jobs(n);
  std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[n]);
  for (size_t i = 0; i < n; ++i) done[i].store(0, std::memory_order_relaxed);
  std::atomic<size_t> next_job{0};
  const unsigned hw = host_threads();
  const unsigned n_workers = std::max(1u, std::min<unsigned>(hw > 2 ? hw - 1 : hw, (unsigned)n));
  const BatchTrace trace;
  auto worker = [&]() {
    for (size_t i; (i = next_job.fetch_add(1)) < n;) {
      if (stdins[i]) {
        try {
          ExecOptions o;
          o.keccak_mode = (KeccakMode)ctx->params.keccak_mode;
          jobs[i].rec = execute(pk->elf, stdins[i]->entries, o);
          stdins[i]->entries.clear();  // consumed, as SP1Stdin is by prove()
        } catch (...) {  // an exception leaving a thread would terminate the process
          jobs[i].rec = ExecutionRecord();
          jobs[i].rec.error = "out of memory while executing the guest";
        }
      }
      done[i].store(1, std::memory_order_release);
    }
  };
  // Joins the workers on every way out of this function (normal return, early return, exception):
  // a joinable std::thread destroyed unjoined calls std::terminate, which would abort across the C ABI.
  struct WorkerGuard {
    std::vector<std::thread> th;
    std::atomic<size_t>* next;
    size_t n;
    ~WorkerGuard() {
      next->store(n);  // no further jobs are claimed
      for (auto& t : th)
        if (t.joinable()) t.join();
    }
  } guard{{}, &next_job, n};
  try {
    for (unsigned t = 0; t < n_workers; ++t) guard.th.emplace_back(worker);
  } catch (...) {
    return ctx->fail(ZKSP_ERR_HIP, "prove: could not start the executor threads");
  }

  // ---- stages 2 and 3, driven from this thread ----
  const size_t max_batch = ctx->params.max_batch;
  // The first wave is small so that the GPU starts while the executor workers are still
  // running ahead; later waves are as large as the workspace allows (larger groups use the
  // GPU better, and the executor is several waves ahead by then).
  const size_t first_wave = n <= 32 ? n : std::max<size_t>(16, std::min(n, max_batch / 4));
  std::string first_err;
  int rc_all = ZKSP_OK;
  Group pending;          // proven (or being proven) on the GPU, not yet assembled
  bool have_pending = false;
  int next_slot = 0;

  auto assemble_group = [&](const Group& g) {
    const uint32_t* bodies = ctx->h_stage2[g.slot];
    parallel_for(g.idxs.size(), 8, [&](size_t j) {
      const size_t i = g.idxs[j];
      const ExecutionRecord& r = jobs[i].rec;
      zksp_proof* p = new (std::nothrow) zksp_proof();
      if (!p) { status[i] = ZKSP_ERR_INVALID_ARG; return; }
      const uint32_t pv_len = (uint32_t)r.public_values.size();
      const size_t hwords = proof_header_words(pv_len, g.np[j]);
      std::vector<uint8_t> head(hwords * 4, 0);
      {
        // public I/O list: every permutation's input state and keccak-f of it
        uint8_t* io = head.data() + (30 + (pv_len + 3) / 4) * 4;
        for (size_t p = 0; p < r.keccak_events.size(); ++p) {
          uint64_t st[25];
          memcpy(st, r.keccak_events[p].state_in, 200);
          memcpy(io + p * 400, st, 200);
          keccak_f1600(st);
          memcpy(io + p * 400 + 200, st, 200);
        }
      }
      uint32_t* w = reinterpret_cast<uint32_t*>(head.data());
      w[0] = kProofMagic; w[1] = kProofVersion; w[2] = (uint32_t)g.logh; w[3] = g.np[j]; w[4] = r.exit_code; w[5] = pv_len;
      memcpy(w + 6, r.pv_digest.data(), 32);
      memcpy(w + 14, r.deferred_digest.data(), 32);
      memcpy(w + 22, pk->vk_digest, 32);
      if (pv_len) memcpy(head.data() + 120, r.public_values.data(), pv_len);
      p->bytes.reserve((hwords + g.bw) * 4);
      p->bytes.insert(p->bytes.end(), head.begin(), head.end());
      const uint8_t* body = reinterpret_cast<const uint8_t*>(bodies + j * g.bw);
      p->bytes.insert(p->bytes.end(), body, body + g.bw * 4);
      std::string perr;
      p->version = kProofVersion;
      if (!parse_proof_header(p->bytes.data(), p->bytes.size(), &p->hdr, &perr)) {
        delete p;
        status[i] = ZKSP_ERR_PROOF_FORMAT;
        return;
      }
      out[i] = p;
      status[i] = ZKSP_OK;
    });
  };
  auto fail_group = [&](const Group& g, int rc) {
    for (size_t i : g.idxs) status[i] = rc;
    rc_all = rc;
  };

  ctx->body_free = nullptr;
  ctx->batch_hint = (int)std::min(max_batch, n);

  for (size_t w0 = 0, wave = first_wave; w0 < n; w0 += wave, wave = max_batch) {
    const size_t w1 = std::min(n, w0 + wave);
    for (size_t i = w0; i < w1; ++i)
      while (!done[i].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    trace.mark("wave executed, first index", w0);
    std::map<int, std::vector<size_t>> by_height;
    for (size_t i = w0; i < w1; ++i) {
      ExecutionRecord& r = jobs[i].rec;
      if (!stdins[i]) continue;
      if (!r.error.empty() || !r.halted) {
        status[i] = ZKSP_ERR_EXECUTOR;
        if (first_err.empty()) first_err = "executor: " + (r.error.empty() ? std::string("guest did not halt") : r.error);
        continue;
      }
      if (r.exit_code != 0) {
        status[i] = ZKSP_ERR_GUEST_PANIC;
        if (first_err.empty()) first_err = "guest panicked (exit code " + std::to_string(r.exit_code) + "): " + r.stderr_text;
        continue;
      }
      if (r.keccak_events.empty()) {
        status[i] = ZKSP_ERR_EXECUTOR;
        if (first_err.empty()) first_err = "guest made no keccak-f calls";
        continue;
      }
      jobs[i].logh = trace_log_height(r.keccak_events.size());
      if (jobs[i].logh > 14) { status[i] = ZKSP_ERR_UNSUPPORTED; continue; }
      by_height[jobs[i].logh].push_back(i);
    }
    for (auto& kv : by_height) {
      for (size_t off = 0; off < kv.second.size(); off += max_batch) {
        Group g;
        g.logh = kv.first;
        const size_t cnt = std::min(max_batch, kv.second.size() - off);
        g.idxs.assign(kv.second.begin() + off, kv.second.begin() + off + cnt);
        g.bw = proof_body_words(g.logh, ctx->params.num_queries);
        g.slot = next_slot;
        size_t max_perms = 0;
        for (size_t i : g.idxs) max_perms = std::max(max_perms, jobs[i].rec.keccak_events.size());
        std::vector<uint64_t> states(cnt * max_perms * 25, 0);
        std::vector<uint32_t> obs(cnt * kInitObs);
        g.np.resize(cnt);
        for (size_t j = 0; j < cnt; ++j) {
          const ExecutionRecord& r = jobs[g.idxs[j]].rec;
          for (size_t p = 0; p < r.keccak_events.size(); ++p)
            memcpy(&states[(j * max_perms + p) * 25], r.keccak_events[p].state_in, 200);
          g.np[j] = (uint32_t)r.keccak_events.size();
          uint32_t* o = &obs[j * kInitObs];
          memcpy(o, pk->vk_digest, 32);
          o[8] = (uint32_t)g.logh;
          o[9] = g.np[j];
          o[10] = r.exit_code & 0xffff;
          o[11] = r.exit_code >> 16;
          for (int k = 0; k < 8; ++k) {
            o[12 + 2 * k] = r.pv_digest[k] & 0xffff;
            o[13 + 2 * k] = r.pv_digest[k] >> 16;
            o[28 + 2 * k] = r.deferred_digest[k] & 0xffff;
            o[29 + 2 * k] = r.deferred_digest[k] >> 16;
          }
        }
        // load_batch synchronises the proving stream, so the previous pass is complete when it
        // returns; that pass's device-to-host copy runs on the copy stream and may still be in
        // flight.  If this group makes the workspace reallocate (another trace height, a larger
        // group) the copy's source would be freed under it: wait for the copy first.
        {
          const Workspace* w = ctx->ws.get();
          const int want = std::max((int)cnt, ctx->batch_hint);
          const bool realloc = !w || w->logh != g.logh || want > w->batch || (int)max_perms > w->max_perms;
          if (realloc && ctx->body_free && hipEventSynchronize(ctx->body_free) != hipSuccess) {
            fail_group(g, ctx->fail(ZKSP_ERR_HIP, "prove: waiting for the device-to-host copy failed"));
            continue;
          }
          if (realloc) ctx->body_free = nullptr;
        }
        int rc = zksp_hip_load_batch(c, g.logh, cnt, max_perms, states.data(), g.np.data(), obs.data());
        trace.mark("previous group finished on the GPU; loaded group of", cnt);
        if (rc == ZKSP_OK) rc = zksp_hip_prove_resident(c);
        if (rc == ZKSP_OK && ctx->h_stage2_words[g.slot] < cnt * g.bw) {
          if (ctx->h_stage2[g.slot]) (void)hipHostFree(ctx->h_stage2[g.slot]);
          ctx->h_stage2[g.slot] = nullptr;
          ctx->h_stage2_words[g.slot] = 0;
          if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage2[g.slot]), cnt * g.bw * 4, hipHostMallocDefault) == hipSuccess)
            ctx->h_stage2_words[g.slot] = cnt * g.bw;
          else
            rc = ctx->fail(ZKSP_ERR_HIP, "prove: pinned staging allocation failed");
        }
        // the copy runs on its own stream behind this pass; the NEXT pass waits for it only before
        // its assemble kernel (Context::body_free), so it overlaps that pass's other kernels
        if (rc == ZKSP_OK &&
            (hipEventRecord(ctx->ev_proved[g.slot], ctx->stream) != hipSuccess ||
             hipStreamWaitEvent(ctx->copy_stream, ctx->ev_proved[g.slot], 0) != hipSuccess ||
             hipMemcpyAsync(ctx->h_stage2[g.slot], ctx->ws->body, cnt * g.bw * 4, hipMemcpyDeviceToHost,
                            ctx->copy_stream) != hipSuccess ||
             hipEventRecord(ctx->ev_copied[g.slot], ctx->copy_stream) != hipSuccess))
          rc = ctx->fail(ZKSP_ERR_HIP, "prove: device-to-host copy failed");
        if (rc == ZKSP_OK) ctx->body_free = ctx->ev_copied[g.slot];
        // while the GPU works on this group, build the previous group's proof objects
        if (have_pending) {
          if (hipEventSynchronize(ctx->ev_copied[pending.slot]) != hipSuccess)
            fail_group(pending, ctx->fail(ZKSP_ERR_HIP, "prove: waiting for the device-to-host copy failed"));
          else
          assemble_group(pending);
          trace.mark("assembled group of", pending.idxs.size());
          have_pending = false;
        }
        if (rc != ZKSP_OK) {
          fail_group(g, rc);
          continue;
        }
        pending = std::move(g);
        have_pending = true;
        next_slot ^= 1;
      }
    }
  }
  if (have_pending) {
    if (hipEventSynchronize(ctx->ev_copied[pending.slot]) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
      fail_group(pending, ctx->fail(ZKSP_ERR_HIP, "prove: stream sync failed"));
    else {
      trace.mark("last group finished on the GPU, size", pending.idxs.size());
      assemble_group(pending);
    }
  }
  ctx->body_free = nullptr;  // every copy has completed: later passes on this client need no wait
  ctx->batch_hint = 0;
  trace.mark("all assembled", n);
  if (!first_err.empty() && rc_all == ZKSP_OK) ctx->error = first_err;
  return rc_all;
#endif
}

int zksp_prove_batch(zksp_client* c, const zksp_pk* pk, zksp_stdin* const* stdins, size_t n, zksp_proof** out,
                     int32_t* status) {
  // nothing unwinds past extern "C": allocation failures inside the pipeline become an error code
  try {
    return prove_batch_impl(c, pk, stdins, n, out, status);
  } catch (const std::exception& e) {
    return c ? c->ctx.fail(ZKSP_ERR_INVALID_ARG, std::string("prove: ") + e.what()) : ZKSP_ERR_INVALID_ARG;
  } catch (...) {
    return c ? c->ctx.fail(ZKSP_ERR_INVALID_ARG, "prove: unknown exception") : ZKSP_ERR_INVALID_ARG;
  }
}

int zksp_prove(zksp_client* c, const zksp_pk* pk, zksp_stdin* stdin_, zksp_proof** out) {
  if (!c || !pk || !stdin_ || !out) return ZKSP_ERR_INVALID_ARG;
  int32_t st = 0;
  zksp_stdin* arr[1] = {stdin_};
  int rc = zksp_prove_batch(c, pk, arr, 1, out, &st);
  return rc != ZKSP_OK ? rc : st;
}

}  // extern "C"
