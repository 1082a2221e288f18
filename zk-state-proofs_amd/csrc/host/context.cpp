#include "context.hpp"

#include "mprover.hpp"  // MachineWorkspace, PrepDevice
#ifdef ZKSP_COMPONENT
#include "prover.hpp"  // Workspace must be complete for ~Context
#endif

#include <cstring>

namespace zksp {

Context::Context() = default;

Context::~Context() {
  if (cleanup.joinable()) cleanup.join();
  if (device >= 0) {
    (void)hipSetDevice(device);
    for (auto& kv : domains) {
      DeviceDomain& d = kv.second;
      uint32_t* ptrs[] = {d.tw_fwd, d.tw_inv, d.twc_fwd, d.twc_inv, d.in_scale_br, d.out_scale_br,
                          d.xs, d.sel_first, d.sel_trans, d.sel_last, d.zh_inv};
      for (uint32_t* p : ptrs)
        if (p) (void)hipFree(p);
    }
    for (auto& kv : qscale)
      for (uint32_t* p : kv.second)
        if (p) (void)hipFree(p);
    if (d_consts) (void)hipFree(d_consts);
    for (uint32_t* hs : h_stage2)
      if (hs) (void)hipHostFree(hs);
    for (int k = 0; k < kH2dBuffers; ++k) {
      if (h2d_ev[k]) { (void)hipEventSynchronize(h2d_ev[k]); (void)hipEventDestroy(h2d_ev[k]); }
      if (h2d_buf[k]) (void)hipHostFree(h2d_buf[k]);
    }
    for (auto& e : event_pool) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    for (hipEvent_t e : ev_proved)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_copied)
      if (e) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    for (hipStream_t sd : side)
      if (sd) (void)hipStreamDestroy(sd);
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    for (hipEvent_t e : join_ev)
      if (e) (void)hipEventDestroy(e);
    if (timer_a) (void)hipEventDestroy(timer_a);
    if (timer_b) (void)hipEventDestroy(timer_b);
#ifdef ZKSP_COMPONENT
    ws.reset();
#endif
    mws.reset();
    retired.clear();
    if (arena) (void)hipFree(arena);
    for (void* p : rec_arena)
      if (p) (void)hipFree(p);
    for (auto& kv : prep)
      for (void* p : kv.second->allocs)
        if (p) (void)hipFree(p);
    prep.clear();
    for (void* p : d_inter)
      if (p) (void)hipFree(p);
    if (stream) (void)hipStreamDestroy(stream);
  }
}

static uint32_t* upload(Context* ctx, const uint32_t* src, size_t words, bool* ok) {
  uint32_t* p = nullptr;
  if (hipMalloc(&p, words * 4) != hipSuccess) {
    *ok = false;
    return nullptr;
  }
  if (hipMemcpyAsync(p, src, words * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) *ok = false;
  return p;
}

const DeviceDomain* Context::domain(int logh) {
  auto it = domains.find(logh);
  if (it != domains.end()) return &it->second;
  if (!has_device() || logh < 1 || logh > 22) {
    fail(9, "domain: unsupported log height");
    return nullptr;
  }
  // heights above 2^14 are NTT-only (two-pass path): no constraint selectors
  const bool full = logh <= 14;
  HostDomain hd;
  build_host_domain(logh, &hd, full);
  DeviceDomain dd;
  dd.logh = logh;
  dd.w_h = hd.w_h;
  bool ok = true;
  dd.tw_fwd = upload(this, hd.tw_fwd.data(), hd.tw_fwd.size(), &ok);
  dd.tw_inv = upload(this, hd.tw_inv.data(), hd.tw_inv.size(), &ok);
  dd.twc_fwd = upload(this, hd.twc_fwd.data(), hd.twc_fwd.size(), &ok);
  dd.twc_inv = upload(this, hd.twc_inv.data(), hd.twc_inv.size(), &ok);
  {
    std::vector<uint32_t> all;
    for (int t = 0; t < 3; ++t) all.insert(all.end(), hd.in_scale_br[t].begin(), hd.in_scale_br[t].end());
    dd.in_scale_br = upload(this, all.data(), all.size(), &ok);
    if (hipStreamSynchronize(stream) != hipSuccess) ok = false;
  }
  dd.out_scale_br = upload(this, hd.out_scale_br.data(), hd.out_scale_br.size(), &ok);
  if (full) {
    dd.xs = upload(this, hd.xs.data(), hd.xs.size(), &ok);
    dd.sel_first = upload(this, hd.sel_first.data(), hd.sel_first.size(), &ok);
    dd.sel_trans = upload(this, hd.sel_trans.data(), hd.sel_trans.size(), &ok);
    dd.sel_last = upload(this, hd.sel_last.data(), hd.sel_last.size(), &ok);
  }
  dd.zh_inv = upload(this, hd.zh_inv, 2, &ok);
  // (shift_k * w_{2Hk}^c)^-1 for every fold round
  Fp shift = Fp::from_canonical(kGen);
  for (int k = 0; k < logh; ++k) {
    int loghk = logh - k;
    Fp w2 = fp_root_of_unity(loghk + 1);
    dd.fold_xinv.push_back(shift.inv().v);
    dd.fold_xinv.push_back((shift * w2).inv().v);
    shift = shift * shift;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) ok = false;  // host vectors go out of scope
  if (!ok) {
    fail(3, "domain: table upload failed");
    return nullptr;
  }
  auto res = domains.emplace(logh, dd);
  return &res.first->second;
}

ProfileSpan::ProfileSpan(Context* c, const char* name) : ctx(c) {
  if (!c->profile) return;
  if (c->event_used == c->event_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    c->event_pool.emplace_back(a, b);
  }
  auto& ev = c->event_pool[c->event_used++];
  idx = c->spans.size();
  c->spans.push_back({name, ev.first, ev.second});
  (void)hipEventRecord(ev.first, c->stream);
}
ProfileSpan::~ProfileSpan() {
  if (idx != (size_t)-1) (void)hipEventRecord(ctx->spans[idx].b, ctx->stream);
}

hipError_t Context::h2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
  // small copies: the runtime stages them itself (it page-locks in place only from a size on)
  if (bytes < ((size_t)64 << 10)) return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
  for (int k = 0; k < kH2dBuffers; ++k) {
    if (h2d_buf[k]) continue;
    if (hipHostMalloc(&h2d_buf[k], kH2dBytes, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h2d_ev[k], hipEventDisableTiming) != hipSuccess) {
      // no page-locked memory to be had: the runtime's own path
      if (h2d_buf[k]) { (void)hipHostFree(h2d_buf[k]); h2d_buf[k] = nullptr; }
      return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    }
  }
  const uint8_t* from = static_cast<const uint8_t*>(src);
  uint8_t* to = static_cast<uint8_t*>(dst);
  for (size_t off = 0; off < bytes; off += kH2dBytes) {
    const size_t len = std::min(kH2dBytes, bytes - off);
    const int k = h2d_next;
    h2d_next = (h2d_next + 1) % kH2dBuffers;
    if (h2d_busy[k]) {  // the transfer that last used this buffer
      const hipError_t e = hipEventSynchronize(h2d_ev[k]);
      if (e != hipSuccess) return e;
    }
    memcpy(h2d_buf[k], from + off, len);
    hipError_t e = hipMemcpyAsync(to + off, h2d_buf[k], len, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipEventRecord(h2d_ev[k], s);
    if (e != hipSuccess) return e;
    h2d_busy[k] = true;
  }
  return hipSuccess;
}

}  // namespace zksp
