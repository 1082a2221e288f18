// Per-client state: HIP device + stream, Poseidon2 constants, per-height domain
// tables and the batch workspace of the device prover.
#pragma once
#include <hip/hip_runtime.h>

#include <thread>

#include <array>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../device/air_machine.hpp"
#include "../device/kernels.h"
#include "executor.hpp"

namespace zksp {

// Builds this repository's Poseidon2 instance (DESIGN.md "Poseidon2 instance"):
// SHAKE256("zksp/poseidon2/babybear/w16/v1") -> 141 rejection-sampled words.
void build_p2_consts(P2Consts* out_monty);
const P2Consts& host_p2_consts();

struct HostDomain {
  int logh = 0;
  std::vector<uint32_t> tw_fwd, tw_inv;          // [H/2] powers of w_H, w_H^-1
  std::vector<uint32_t> twc_fwd, twc_inv;        // [H] per-stage: [2^(t-1) + j] = w_{2^t}^(+-j), j < 2^(t-1)
  std::vector<uint32_t> in_scale_br[3];          // [H]: trace (shift 1), quotient chunk 0, chunk 1
  std::vector<uint32_t> out_scale_br;            // [2][H]
  std::vector<uint32_t> xs, sel_first, sel_trans, sel_last;  // [2][H]
  uint32_t zh_inv[2];
  uint32_t w_h;
};
void build_host_domain(int logh, HostDomain* d, bool full = true);

struct DeviceDomain {
  int logh = 0;
  uint32_t *tw_fwd = nullptr, *tw_inv = nullptr, *twc_fwd = nullptr, *twc_inv = nullptr, *in_scale_br = nullptr /*[3][H]: shift 1, g, g*w_2H*/,
           *out_scale_br = nullptr, *xs = nullptr, *sel_first = nullptr, *sel_trans = nullptr, *sel_last = nullptr,
           *zh_inv = nullptr;
  uint32_t w_h = 0;
  std::vector<uint32_t> fold_xinv;  // [logh][2] host copies of (shift_k * w_{2Hk}^c)^-1
};

struct Params {
  uint32_t num_queries = 100;
  uint32_t pow_bits = 16;
  uint32_t max_batch = 192;  // an upper bound: a chunk is also capped by the free HBM and by half of the call
  int keccak_mode = 2;
  int proof_mode = 1;  // ZKSP_PROOF_MACHINE
};

#ifdef ZKSP_COMPONENT
struct Workspace;  // prover.cpp: the round-1 keccak-chip component path (include/zksp_component.h)
#endif
struct MachineWorkspace;  // mprover.cpp
struct PrepDevice;

struct Context {
  int device = -1;
  hipStream_t stream = nullptr;
  P2Consts* d_consts = nullptr;
  Params params;
  std::map<int, DeviceDomain> domains;
  std::map<uint32_t, std::vector<uint32_t*>> qscale;  // custom in_shift scale tables for zksp_hip_lde
#ifdef ZKSP_COMPONENT
  std::unique_ptr<Workspace> ws;
#endif
  // machine proof: batch workspace, per-program preprocessed tables (by verifying-key digest),
  // device copies of the chips' bus interactions
  std::unique_ptr<MachineWorkspace> mws;
  void* arena = nullptr;   // the machine workspace's device memory: laid out again per shape, grown when too small
  size_t arena_bytes = 0;
  // The executor's records of a batch live OUTSIDE that arena, in two buffers of their own (the resident set and the spare
  // set, MachineWorkspace::rec_slot says which is which): a chunk of ANOTHER shape can then be uploaded while the current
  // one is proven, and laying the arena out for the new shape moves nothing that an upload in flight writes.
  void* rec_arena[2] = {nullptr, nullptr};
  size_t rec_bytes[2] = {0, 0};
  // workspaces replaced while passes enqueued from them may still read their host-side tables: buried when the stream is idle
  std::vector<std::unique_ptr<MachineWorkspace>> retired;
  int spare_batch_hint = 0;  // prove_batch: the size of the group the spare set's chunk belongs to
  std::map<std::array<uint32_t, 8>, std::unique_ptr<PrepDevice>> prep;
  void* d_inter[mach::kNumChips] = {nullptr};  // indexed by chip
  uint32_t* h_stage2[2] = {nullptr, nullptr};  // pinned host staging for proof bodies (double buffered)
  size_t h_stage2_words[2] = {0, 0};
  std::vector<uint32_t> h_stage2_pageable[2];  // ... pageable stand-ins while no pinned memory of that size can be had
  // prove_batch copies a group's bodies to the host on its own stream while the next group is
  // proven: ev_proved[slot] marks the end of a pass, ev_copied[slot] the end of its copy.  The next
  // pass waits for body_free (the last copy) only right before its assemble kernel, the one launch
  // that overwrites the body buffer.
  // Small batches (latency): the chips of a stage are independent, and one proof's kernels are far too small to fill the
  // GPU one after the other - at batch <= kSideMaxBatch the per-chip loops of a pass fan out over the side streams, by
  // chip height, and join again at the end of the stage (mprover.cpp StageFork).
  // (two side streams: the runtime maps streams onto four hardware queues by default, and the copy stream wants one)
  // (kSideMaxBatch: measured at 32, 64 and 128 too - resident passes of 9 .. 48 proofs gain 2-19 % from the lanes, the
  // crossover with the one-lane path is near 64 - but a prove_batch call of SEVERAL chunks, whose chunks are uploaded and
  // fetched on the copy stream while the one before is proven, LOSES: slot-d5x256 436 -> 350 proofs/s, acct-d8x1024 304 ->
  // 288 at 64; 435 -> 411 and 308 -> 298 at 48, with GPU_MAX_HW_QUEUES=8 as with the default 4 - it is not a shared hardware
  // queue.  So the lanes carry batches of at most 48 proofs when nothing is copied beside the pass (resident passes, a single
  // proof, a prove_batch call of one chunk) and of at most eight inside a pipelined prove_batch (`pipelined`; the first two
  // waves of a call that ramps up are exempt: api_prove.cpp): round 4 had eight everywhere, which made a caller with 9 .. 15
  // runs slower than one with 8.)
  static constexpr int kSideStreams = 2, kSideMaxBatch = 48, kSidePipelinedMaxBatch = 8;
  bool pipelined = false;  // set by prove_batch while chunks are uploaded / fetched beside the passes
  int lane_max_batch() const { return pipelined ? kSidePipelinedMaxBatch : kSideMaxBatch; }
  hipStream_t side[kSideStreams] = {nullptr, nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[kSideStreams] = {nullptr, nullptr};
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_proved[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
  hipEvent_t body_free = nullptr;  // not owned: one of ev_copied, or null
  int batch_hint = 0;  // prove_batch: the largest group it will load, so the workspace is sized once
  // Host-to-device copies of the executor's records go through page-locked staging buffers of the client's own
  // (Context::h2d): handed pageable memory, the runtime page-locks the caller's pages for the transfer and unlocks them
  // afterwards, and a kernel that runs meanwhile stalls for tens of milliseconds on the page-table update (one 30 ms stall
  // per uploaded chunk in a prove_batch call: rocprof showed a 0.3 ms kernel taking 31.7 ms; with the runtime told never
  // to lock in place, GPU_PINNED_MIN_XFER_SIZE, a call of 1 024 runs took 2 990 ms instead of 3 055-3 097).
  static constexpr int kH2dBuffers = 4;
  static constexpr size_t kH2dBytes = (size_t)8 << 20;
  void* h2d_buf[kH2dBuffers] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t h2d_ev[kH2dBuffers] = {nullptr, nullptr, nullptr, nullptr};
  bool h2d_busy[kH2dBuffers] = {false, false, false, false};
  int h2d_next = 0;
  // dst (device) <- src (any host memory), enqueued on `s`; src may be freed when this returns.  hipSuccess, or the error.
  // (called by the thread that drives the client - a client is not re-entrant; a buffer of the ring is reused once the
  // transfer that last read it has completed, whichever stream it was on)
  hipError_t h2d(void* dst, const void* src, size_t bytes, hipStream_t s);
  std::string error;
  // profiling
  bool profile = false;
  struct Span { std::string name; hipEvent_t a, b; };
  std::vector<Span> spans;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
  size_t event_used = 0;
  hipEvent_t timer_a = nullptr, timer_b = nullptr;

  Context();  // out of line: members hold types that are incomplete here
  std::thread cleanup;  // frees the last prove_batch call's execution records off the caller's clock
  ~Context();
  bool has_device() const { return device >= 0; }
  int fail(int code, const std::string& msg) { error = msg; return code; }
  const DeviceDomain* domain(int logh);  // builds + uploads on first use; nullptr on error
};

constexpr uint32_t kProofMagic = 0x50534B5Au;  // "ZKSP": the first word of every proof

// RAII span used when ctx->profile is on
struct ProfileSpan {
  Context* ctx;
  size_t idx = (size_t)-1;
  ProfileSpan(Context* c, const char* name);
  ~ProfileSpan();
};

#ifdef ZKSP_COMPONENT
size_t proof_body_words(int logh, uint32_t num_queries);
size_t proof_header_words(uint32_t pv_len, uint32_t n_perms);
int bus_io_log_rows(int logh);
#endif

#define ZKSP_HIP_CHECK(ctx, call)                                                         \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return (ctx)->fail(3 /*ZKSP_ERR_HIP*/, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

}  // namespace zksp
