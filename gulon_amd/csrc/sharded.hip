// Row-sharded PQIndex over the GPUs of one node behind the C ABI: ONE host process (the JVM), one
// stream + workspace per shard, RCCL all-gathers over xGMI between the devices.
//
// The code matrix is cut into contiguous row ranges -- the from/until contract of PQIndex.batchQuery
// (Index.scala:417-419); every shard scans its rows and the per-shard partial top-(K+1) lists are
// merged with TopKHeap.merge's rule (TopKHeap.scala:44-53, as Index.scala:279 uses it) under the
// deterministic (distance, row id) order, so results do not depend on the number of shards.  The three
// exchanges are the ones gulon_amd/sharded.py performs between processes:
//   1. [B][K+1] sample bounds of every shard          (gulon_index_scan_bounds_dev)
//   2. [2][B][K+1] partial lists (distance bits, ids) (gulon_index_scan_partial_bounded_dev)
//   3. candidate packs of the tie-flagged queries     (gulon_index_replay_collect_dev), in rounds until
//      every flagged query has been replayed with the literal TopKHeap (TopKHeap.scala:57-79)
// each one ncclAllGather per device inside one ncclGroupStart/End.  Several shards may share a device
// (devices[] may repeat): a device then contributes several slots to each gather.
//
// RCCL is loaded with dlopen at the first gulon_sharded_index_create: libgulon_hip.so itself carries no
// dependency on it (processes that bring their own collectives -- bench.py under torch.distributed --
// never load a second copy).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <map>

#include "scan.hpp"

using namespace gulon;

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string where;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {getenv("GULON_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
      if (!nm || !*nm) continue;
      r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) { r.where = nm; break; }
    }
    if (!r.handle) return;
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name))
    SYM(GetVersion, "ncclGetVersion");
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    if (!r.GetVersion || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
      dlclose(r.handle);
      r.handle = nullptr;
    }
  });
  return r;
}

#define NCCL_CHECK(expr)                                                                              \
  do {                                                                                                \
    ncclResult_t _r = (expr);                                                                         \
    if (_r != ncclSuccess) {                                                                          \
      ::gulon::set_error("%s failed: %s (%s:%d)", #expr,                                              \
                         rccl().GetErrorString ? rccl().GetErrorString(_r) : "?", __FILE__, __LINE__); \
      throw ::gulon::DeviceError{GULON_ERR_DEVICE};                                                   \
    }                                                                                                 \
  } while (0)

struct DeviceGuard {   // hipSetDevice is per-thread state: put the caller's device back
  int prev = 0;
  DeviceGuard() { (void)hipGetDevice(&prev); }
  ~DeviceGuard() { (void)hipSetDevice(prev); }
};

struct Slot {             // one shard (or a padding slot: no rows) on a device
  gulon_index *ix = nullptr;   // the shard's index (owned) or a context over a local shard (padding)
  bool padding = false;
  int rows = 0;
};

struct Dev {
  int device = 0;
  hipStream_t st = nullptr;
  ncclComm_t comm = nullptr;
  std::vector<Slot> slots;    // exactly `spd` of them
  DevBuf<float> q, bd_send, bd_all, od;
  DevBuf<int> pk_send, pk_all, rp_send, rp_all, oi, oc, of;
};

__global__ void fill_pair(float *__restrict__ v, int *__restrict__ i, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) { v[t] = INFINITY; i[t] = INT_MAX; }
}

}  // namespace

struct gulon_sharded_index {
  int n = 0, d = 0, m = 0, k = 0, shards = 0, spd = 1;
  std::vector<Dev> devs;
  std::vector<int> lo, hi;
  int rccl_version = 0;
  int last_rounds = 0, last_flagged = 0;
  std::mutex mu;
  ~gulon_sharded_index() {
    DeviceGuard g;
    for (auto &dv : devs) {
      (void)hipSetDevice(dv.device);
      if (dv.st) (void)hipStreamSynchronize(dv.st);
      // padding slots borrow the buffers of the device's owner slot: they go first
      for (auto &s : dv.slots)
        if (s.ix && s.padding) { delete s.ix; s.ix = nullptr; }
      for (auto &s : dv.slots)
        if (s.ix) gulon_index_destroy(s.ix);
      dv.slots.clear();
      if (dv.comm && rccl().CommDestroy) (void)rccl().CommDestroy(dv.comm);
      dv.q.release(); dv.bd_send.release(); dv.bd_all.release(); dv.od.release();
      dv.pk_send.release(); dv.pk_all.release(); dv.rp_send.release(); dv.rp_all.release();
      dv.oi.release(); dv.oc.release(); dv.of.release();
      if (dv.st) (void)hipStreamDestroy(dv.st);
    }
  }
};

namespace {

// one ncclAllGather per device, grouped: `bufs(dv)` = (send, recv); `count` elements per device
template <class Bufs>
void all_gather(gulon_sharded_index *sx, Bufs bufs, size_t count, ncclDataType_t ty) {
  Rccl &r = rccl();
  // the group is closed on EVERY path: a failure between GroupStart and GroupEnd that left it open would nest
  // every later collective of this thread inside it -- they would never launch
  struct Group {
    Rccl &r;
    bool open = false;
    ~Group() { if (open) (void)r.GroupEnd(); }
  } group{r};
  NCCL_CHECK(r.GroupStart());
  group.open = true;
  for (auto &dv : sx->devs) {
    HIP_CHECK(hipSetDevice(dv.device));
    std::pair<const void *, void *> sr = bufs(dv);
    NCCL_CHECK(r.AllGather(sr.first, sr.second, count, ty, dv.comm, dv.st));
  }
  group.open = false;
  NCCL_CHECK(r.GroupEnd());
}

// rows [lo, hi) of an [m][bytesPerCode(n)] EncodedMatrix as an [m][bytesPerCode(hi - lo)] one
std::vector<uint8_t> slice_codes(const uint8_t *codes, int n, int m, int width, int lo, int hi) {
  int bpc_all = 0, bpc = 0;
  gulon_coder_bytes(width, n, &bpc_all);
  gulon_coder_bytes(width, hi - lo, &bpc);
  std::vector<uint8_t> out((size_t)m * bpc + 1);
  if (width == 8) {
    for (int j = 0; j < m; j++) memcpy(out.data() + (size_t)j * bpc, codes + (size_t)j * bpc_all + lo, (size_t)(hi - lo));
    return out;
  }
  if (width == 0 || hi == lo) return out;
  std::vector<int32_t> idx((size_t)n);
  for (int j = 0; j < m; j++) {
    GULON_REQUIRE(gulon_coder_unpack(width, codes + (size_t)j * bpc_all, n, idx.data()) == GULON_OK, "bad codes");
    GULON_REQUIRE(gulon_coder_build(width, idx.data() + lo, hi - lo, out.data() + (size_t)j * bpc) == GULON_OK, "bad codes");
  }
  return out;
}

}  // namespace

GULON_API int32_t gulon_sharded_index_create(const uint8_t *codes, int32_t n, int32_t d, int32_t m, int32_t k,
                                             const float *cents, const int32_t *devices, int32_t n_shards,
                                             gulon_sharded_index **out) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    GULON_REQUIRE(n >= 0 && d >= 1 && m >= 1 && m <= d && k >= 1, "bad index shape n=%d d=%d m=%d k=%d", n, d, m, k);
    GULON_REQUIRE(cents != nullptr && (codes != nullptr || n == 0) && devices != nullptr, "null input");
    GULON_REQUIRE(n_shards >= 1 && n_shards <= 64, "n_shards must be in [1, 64]");
    int ndev_visible = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev_visible));
    for (int s = 0; s < n_shards; s++)
      GULON_REQUIRE(devices[s] >= 0 && devices[s] < ndev_visible, "shard %d: device %d of %d", s, devices[s], ndev_visible);
    int width = -1;
    GULON_REQUIRE(gulon_coder_width(k, &width) == GULON_OK && width >= 0, "too many clusters: %d", k);  // PQ.scala:12-15
    Rccl &r = rccl();
    GULON_UNSUPPORTED(!r.handle, "librccl.so could not be loaded (set GULON_RCCL_LIB): the sharded index needs RCCL");

    DeviceGuard guard;
    std::unique_ptr<gulon_sharded_index> sx(new gulon_sharded_index());
    sx->n = n; sx->d = d; sx->m = m; sx->k = k; sx->shards = n_shards;
    NCCL_CHECK(r.GetVersion(&sx->rccl_version));
    // devices in order of first appearance; shard s covers rows [n*s/S, n*(s+1)/S) (sharded.py: shard_bounds)
    std::map<int, int> pos;
    for (int s = 0; s < n_shards; s++) {
      if (!pos.count(devices[s])) {
        pos[devices[s]] = (int)sx->devs.size();
        sx->devs.emplace_back();
        sx->devs.back().device = devices[s];
      }
      sx->lo.push_back((int)((long long)n * s / n_shards));
      sx->hi.push_back((int)((long long)n * (s + 1) / n_shards));
    }
    for (int s = 0; s < n_shards; s++) {
      Dev &dv = sx->devs[pos[devices[s]]];
      HIP_CHECK(hipSetDevice(dv.device));
      std::vector<uint8_t> part = slice_codes(codes, n, m, width, sx->lo[s], sx->hi[s]);
      Slot sl;
      sl.rows = sx->hi[s] - sx->lo[s];
      int32_t rc = gulon_index_create(part.data(), sl.rows, d, m, k, cents, sx->lo[s], &sl.ix);
      if (rc != GULON_OK) throw DeviceError{rc};
      dv.slots.push_back(sl);
    }
    for (auto &dv : sx->devs) sx->spd = std::max(sx->spd, (int)dv.slots.size());
    for (auto &dv : sx->devs) {
      HIP_CHECK(hipSetDevice(dv.device));
      HIP_CHECK(hipStreamCreateWithFlags(&dv.st, hipStreamNonBlocking));
      while ((int)dv.slots.size() < sx->spd) {   // padding slots: an empty row range of a local shard
        Slot sl;
        sl.padding = true;
        sl.ix = make_context(dv.slots[0].ix);
        dv.slots.push_back(sl);
      }
    }
    {
      std::vector<int> devlist;
      for (auto &dv : sx->devs) devlist.push_back(dv.device);
      std::vector<ncclComm_t> comms(devlist.size());
      NCCL_CHECK(r.CommInitAll(comms.data(), (int)devlist.size(), devlist.data()));
      for (size_t i = 0; i < comms.size(); i++) sx->devs[i].comm = comms[i];
    }
    *out = sx.release();
  });
}

GULON_API int32_t gulon_sharded_index_destroy(gulon_sharded_index *idx) {
  return guarded([&] { delete idx; });
}

GULON_API int32_t gulon_sharded_index_info(const gulon_sharded_index *idx, int32_t *n_shards, int32_t *n_devices,
                                           int32_t *rccl_version, int32_t *last_replay_rounds,
                                           int32_t *last_flagged_queries) {
  return guarded([&] {
    GULON_REQUIRE(idx != nullptr, "index is null");
    if (n_shards) *n_shards = idx->shards;
    if (n_devices) *n_devices = (int32_t)idx->devs.size();
    if (rccl_version) *rccl_version = idx->rccl_version;
    if (last_replay_rounds) *last_replay_rounds = idx->last_rounds;
    if (last_flagged_queries) *last_flagged_queries = idx->last_flagged;
  });
}

GULON_API int32_t gulon_sharded_index_batch_query(gulon_sharded_index *sx, const float *queries, int32_t b, int32_t k_nn,
                                                  int32_t *out_idx, float *out_dist, int32_t *out_count,
                                                  int32_t *out_flags) {
  return guarded([&] {
    GULON_REQUIRE(sx != nullptr, "index is null");
    GULON_REQUIRE(b >= 0 && k_nn >= 0, "k and batch size must be non-negative");
    GULON_UNSUPPORTED(k_nn + 1 > GULON_MAX_K_PEELED, "k_nn = %d > %d on a sharded index", k_nn, GULON_MAX_K_PEELED - 1);
    if (b == 0) return;
    GULON_REQUIRE(queries != nullptr && (k_nn == 0 || (out_idx != nullptr && out_dist != nullptr)), "null argument");
    if (k_nn == 0) {
      if (out_count) memset(out_count, 0, sizeof(int32_t) * (size_t)b);
      if (out_flags) memset(out_flags, 0, sizeof(int32_t) * (size_t)b);
      return;
    }
    std::lock_guard<std::mutex> lock(sx->mu);
    DeviceGuard guard;
    const int K = k_nn, keff = K + 1, spd = sx->spd, nD = (int)sx->devs.size(), lists = nD * spd;
    // k_nn beyond a wavefront list (Tests.scala asks for up to 1000 neighbours): every shard peels its K+1 best 64
    // at a time, the lists are merged pairwise; ties keep the (distance, row id) order and their flags (no replay)
    const bool large_k = K > GULON_MAX_K;
    const size_t nb = (size_t)b * keff;             // one [B][K+1] array
    const int F0 = GULON_REPLAY_MAX_FLAGGED, C0 = GULON_REPLAY_POOL;
    const size_t words0 = (size_t)gulon_replay_pack_words(F0, C0);
    const int F1 = 128;                             // later rounds: more queries per exchange
    const size_t words1 = (size_t)gulon_replay_pack_words(F1, C0);
    for (auto &dv : sx->devs) {
      HIP_CHECK(hipSetDevice(dv.device));
      dv.q.ensure((size_t)b * sx->d);
      dv.bd_send.ensure(spd * nb); dv.bd_all.ensure(lists * nb);
      dv.pk_send.ensure(spd * 2 * nb); dv.pk_all.ensure(lists * 2 * nb);
      dv.rp_send.ensure(spd * words0); dv.rp_all.ensure(lists * words0);
      dv.oi.ensure((size_t)b * K); dv.od.ensure((size_t)b * K); dv.oc.ensure(b); dv.of.ensure(b);
      HIP_CHECK(hipMemcpyAsync(dv.q.p, queries, sizeof(float) * (size_t)b * sx->d, hipMemcpyHostToDevice, dv.st));
    }
    auto status = [](int32_t rc) { if (rc != GULON_OK) throw DeviceError{rc}; };
    // 1. sample bounds of every shard -> all-gather
    if (!large_k) {
      for (auto &dv : sx->devs) {
        HIP_CHECK(hipSetDevice(dv.device));
        for (int s = 0; s < spd; s++) {
          Slot &sl = dv.slots[s];
          status(gulon_index_scan_bounds_dev(sl.ix, dv.q.p, b, K, 0, sl.rows, dv.bd_send.p + s * nb, dv.st));
        }
      }
      all_gather(sx, [](Dev &dv) { return std::pair<const void *, void *>(dv.bd_send.p, dv.bd_all.p); }, spd * nb, ncclFloat);
    }
    // 2. the scan against the bound of the union -> all-gather of the packed partial lists -> merge everywhere
    for (auto &dv : sx->devs) {
      HIP_CHECK(hipSetDevice(dv.device));
      for (int s = 0; s < spd; s++) {
        Slot &sl = dv.slots[s];
        int *pk = dv.pk_send.p + (size_t)s * 2 * nb;
        if (sl.rows == 0) {   // no rows: the empty list, without the host round trip of the general path
          hipLaunchKernelGGL(fill_pair, dim3(ceil_div((long long)nb, 256)), dim3(256), 0, dv.st,
                             reinterpret_cast<float *>(pk), pk + nb, (long long)nb);
          HIP_CHECK(hipGetLastError());
          continue;
        }
        if (large_k)
          status(gulon_index_scan_partial_dev(sl.ix, dv.q.p, b, K, 0, sl.rows, reinterpret_cast<float *>(pk), pk + nb, dv.st));
        else
          status(gulon_index_scan_partial_bounded_dev(sl.ix, dv.q.p, b, K, 0, sl.rows, dv.bd_all.p, lists,
                                                      reinterpret_cast<float *>(pk), pk + nb, dv.st));
      }
    }
    all_gather(sx, [](Dev &dv) { return std::pair<const void *, void *>(dv.pk_send.p, dv.pk_all.p); }, spd * 2 * nb,
               ncclInt32);
    for (auto &dv : sx->devs) {
      HIP_CHECK(hipSetDevice(dv.device));
      status(gulon_topk_merge_dev(reinterpret_cast<const float *>(dv.pk_all.p), dv.pk_all.p + nb, lists,
                                  (int64_t)(2 * nb), b, K, dv.oi.p, dv.od.p, dv.oc.p, dv.of.p, dv.st));
    }
    // 3. tie-flagged queries: candidate rows of every shard -> all-gather -> literal TopKHeap on every device;
    //    first round sized for the common case, further rounds (after one look at the flag count) until all are done
    int skip = 0, rounds = 0, flagged = 0;
    while (!large_k) {
      const int F = rounds == 0 ? F0 : F1;
      const size_t words = rounds == 0 ? words0 : words1;
      for (auto &dv : sx->devs) {
        HIP_CHECK(hipSetDevice(dv.device));
        dv.rp_send.ensure(spd * words); dv.rp_all.ensure(lists * words);
        for (int s = 0; s < spd; s++) {
          Slot &sl = dv.slots[s];
          status(gulon_index_replay_collect_dev(sl.ix, dv.q.p, b, K, 0, sl.rows, dv.of.p, skip, F, C0,
                                                dv.rp_send.p + s * words, dv.st));
        }
      }
      all_gather(sx, [](Dev &dv) { return std::pair<const void *, void *>(dv.rp_send.p, dv.rp_all.p); }, spd * words,
                 ncclInt32);
      for (auto &dv : sx->devs) {
        HIP_CHECK(hipSetDevice(dv.device));
        status(gulon_replay_apply_dev(dv.rp_all.p, lists, F, C0, b, K, dv.oi.p, dv.od.p, dv.oc.p, dv.of.p, dv.st));
      }
      rounds++;
      Dev &d0 = sx->devs[0];
      HIP_CHECK(hipSetDevice(d0.device));
      int hdr[4] = {0, 0, 0, 0};
      HIP_CHECK(hipMemcpyAsync(hdr, d0.rp_send.p, sizeof(hdr), hipMemcpyDeviceToHost, d0.st));
      HIP_CHECK(hipStreamSynchronize(d0.st));
      flagged = hdr[3];
      skip += F;
      if (skip >= flagged) break;
    }
    for (auto &dv : sx->devs) {   // all-NaN queries: the reference's first-K-rows answer (TopKHeap.scala:69-79)
      HIP_CHECK(hipSetDevice(dv.device));
      status(gulon_nan_queries_fix_dev(dv.q.p, b, sx->d, K, sx->n, dv.oi.p, dv.od.p, dv.oc.p, dv.of.p, dv.st));
    }
    sx->last_rounds = rounds;
    sx->last_flagged = flagged;
    Dev &d0 = sx->devs[0];
    HIP_CHECK(hipSetDevice(d0.device));
    d0.oi.download(out_idx, (size_t)b * K, d0.st);
    d0.od.download(out_dist, (size_t)b * K, d0.st);
    if (out_count) d0.oc.download(out_count, b, d0.st);
    if (out_flags) d0.of.download(out_flags, b, d0.st);
    for (auto &dv : sx->devs) {
      HIP_CHECK(hipSetDevice(dv.device));
      HIP_CHECK(hipStreamSynchronize(dv.st));
    }
  });
}
