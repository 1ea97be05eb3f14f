// glia_amd/csrc/slab_dist.cpp -- C1 behind the C ABI: the z-slab route of volumes split across GPUs (no reference
// counterpart, GLIA is single-node).  BASELINE.json's north star: "volumes too big for one GPU are slab-partitioned across the 8
// MI355X with RCCL over xGMI exchanging only the cross-slab boundary regions".  The same three steps as glia_amd/slab.py, here in
// the library so that a C / C++ host (cli/merge_order_pb, cli/merge_order_bc --slabs) can run them:
//   1. every rank builds the partial map of its slab (glia_hmt_rag_build_slab), flags the records whose label occurs on a plane
//      next to a cut (rag_cut_flags) and sorts its records by destination: flagged ones to owner = hash(label) mod N (a pair goes
//      with its first label), the others stay "interior";
//   2. the flagged records take a keyed owner exchange -- one unpadded point-to-point transfer per (source, destination, array),
//      all in ONE ncclGroup, so every xGMI link carries traffic at once -- and each owner reduces what it received by key;
//   3. reduced cut records and interior records travel ONCE to the rank that runs the merge loop, which reduces by key a last time
//      (a record the flags missed -- a label in two slabs that never touches a cut plane -- is still combined: the flags decide the
//      route, never the result).
// The transport is a table of transfers {source rank, destination rank, pointers, bytes}: between two ranks of THIS process it is
// a device-to-device copy, otherwise ncclSend / ncclRecv.  A communicator made by glia_hmt_comm_create_local holds all N ranks in
// one process on one GPU (tests, and volumes whose slabs are processed one after the other); one made by glia_hmt_comm_create_rccl
// holds one rank per process.  RCCL is loaded with dlopen when such a communicator is made: the library itself does not link it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstring>
#include <vector>

#include "api_types.hpp"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  int load() {
    if (lib) return GLIA_HMT_OK;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { set_error(std::string("comm_create_rccl: cannot load librccl.so: ") + dlerror()); return GLIA_HMT_ERR_UNSUPPORTED; }
#define GLIA_SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f) { set_error("comm_create_rccl: librccl.so lacks nccl" #f); return GLIA_HMT_ERR_UNSUPPORTED; }
    GLIA_SYM(GetUniqueId) GLIA_SYM(CommInitRank) GLIA_SYM(CommDestroy) GLIA_SYM(GroupStart) GLIA_SYM(GroupEnd) GLIA_SYM(Send) GLIA_SYM(Recv)
    GLIA_SYM(AllGather) GLIA_SYM(GetErrorString)
#undef GLIA_SYM
    return GLIA_HMT_OK;
  }
};
Rccl g_rccl;
#define GLIA_NCCL_TRY(expr)                                                                                 \
  do {                                                                                                      \
    ncclResult_t _r = (expr);                                                                               \
    if (_r != ncclSuccess) { set_error(std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); return GLIA_HMT_ERR_HIP; } \
  } while (0)

}  // namespace

struct glia_hmt_comm {
  glia_hmt_ctx* ctx = nullptr;
  int world = 1;
  std::vector<int> local;        // ranks held by this process, ascending
  ncclComm_t nccl = nullptr;     // one rank per process (RCCL); null for a local communicator
  bool is_local(int r) const { for (int l : local) if (l == r) return true; return false; }
};

namespace {

struct Transfer { int src, dst; const void* from; void* to; size_t bytes; };

// every transfer of one exchange step; entries between two remote ranks are skipped.  All ranks build the same table in the same
// order (pointers are only meaningful where the rank is local), so sends and receives pair up inside the group.
int run_transfers(glia_hmt_comm* cm, const std::vector<Transfer>& ts, uint64_t* bytes_sent) {
  hipStream_t s = cm->ctx->stream;
  if (cm->nccl) GLIA_NCCL_TRY(g_rccl.GroupStart());
  // an error between GroupStart and GroupEnd must not leave the group open (the communicator would be unusable): the loop runs
  // inside a lambda and GroupEnd is called whatever it returns
  const int rc = [&]() -> int {
    for (const Transfer& t : ts) {
      if (!t.bytes) continue;
      const bool ls = cm->is_local(t.src), ld = cm->is_local(t.dst);
      if (ls && ld) { if (t.from != t.to) GLIA_HIP_TRY(hipMemcpyAsync(t.to, t.from, t.bytes, hipMemcpyDeviceToDevice, s)); }
      else if (ls) GLIA_NCCL_TRY(g_rccl.Send(t.from, t.bytes, ncclUint8, t.dst, cm->nccl, s));
      else if (ld) GLIA_NCCL_TRY(g_rccl.Recv(t.to, t.bytes, ncclUint8, t.src, cm->nccl, s));
      if (ls && t.src != t.dst && bytes_sent) *bytes_sent += t.bytes;
    }
    return GLIA_HMT_OK;
  }();
  if (cm->nccl) { if (rc) { (void)g_rccl.GroupEnd(); return rc; } GLIA_NCCL_TRY(g_rccl.GroupEnd()); }
  if (rc) return rc;
  GLIA_HIP_TRY(hipStreamSynchronize(s));
  return GLIA_HMT_OK;
}

// every rank contributes n 64-bit words; all[r * n ..] = rank r's (host arrays)
int all_gather_u64(glia_hmt_comm* cm, const std::vector<std::vector<uint64_t>>& mine /* per local rank */, int n, std::vector<uint64_t>* all) {
  all->assign((size_t)cm->world * n, 0);
  for (size_t i = 0; i < cm->local.size(); ++i) std::copy(mine[i].begin(), mine[i].end(), all->begin() + (size_t)cm->local[i] * n);
  if (!cm->nccl || cm->world == 1) return GLIA_HMT_OK;
  hipStream_t s = cm->ctx->stream;
  uint64_t *d_in = nullptr, *d_out = nullptr;
  GLIA_HIP_TRY(hipMalloc(&d_in, sizeof(uint64_t) * n));
  GLIA_HIP_TRY(hipMalloc(&d_out, sizeof(uint64_t) * n * cm->world));
  GLIA_HIP_TRY(hipMemcpyAsync(d_in, mine[0].data(), sizeof(uint64_t) * n, hipMemcpyHostToDevice, s));
  GLIA_NCCL_TRY(g_rccl.AllGather(d_in, d_out, (size_t)n, ncclUint64, cm->nccl, s));
  GLIA_HIP_TRY(hipMemcpyAsync(all->data(), d_out, sizeof(uint64_t) * n * cm->world, hipMemcpyDeviceToHost, s));
  GLIA_HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(d_in); (void)hipFree(d_out);
  return GLIA_HMT_OK;
}

// ---- records sorted by destination ----------------------------------------------------------------------------------------
__global__ void dest_keys(const uint32_t* label, const uint8_t* cut, uint32_t n, uint32_t world, uint32_t* key, uint32_t* idx, uint32_t* count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // owner of a label: a multiplicative hash, mod the number of ranks (as glia_amd/slab.py owner_of); `world` = stays interior
  const unsigned long long h = (unsigned long long)label[i] * 2654435761ull;
  const uint32_t d = cut[i] ? (uint32_t)(((h >> 11) & 0x7FFFFFFFull) % world) : world;
  key[i] = d; idx[i] = i;
  atomicAdd(&count[d], 1u);
}
__global__ void permute_rows(const uint32_t* src, const uint32_t* idx, uint32_t n, int words, uint32_t* dst) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned long long)n * words) return;
  const uint32_t i = (uint32_t)(t / words); const int w = (int)(t % words);
  dst[t] = src[(size_t)idx[i] * words + w];
}

struct Sorted {                 // one rank's records in destination order (owned device arrays), counts per destination [world + 1]
  RagArrays arr;
  std::vector<uint64_t> rcount, pcount, vcount;      // vcount: boundary values per destination (0 without value runs)
  std::vector<void*> owned;
  void release() { for (void* p : owned) (void)hipFree(p); owned.clear(); }
};

int sort_by_destination(const RagArrays& a, const uint8_t* d_rcut, const uint8_t* d_pcut, int world, hipStream_t s, Sorted* out) {
  out->arr = RagArrays(); out->arr.R = a.R; out->arr.P = a.P; out->arr.K = a.K;
  for (int c = 0; c < kMaxChannels; ++c) out->arr.c_bins[c] = a.c_bins[c];
  for (int kind = 0; kind < 2; ++kind) {
    const uint32_t n = (uint32_t)(kind ? a.P : a.R);
    std::vector<uint64_t>& cnt = kind ? out->pcount : out->rcount;
    cnt.assign((size_t)world + 1, 0);
    auto alloc = [&](uint32_t** p, size_t words) -> int { GLIA_HIP_TRY(hipMalloc(p, sizeof(uint32_t) * (words ? words : 1))); out->owned.push_back(*p); return GLIA_HMT_OK; };
    uint32_t *lab = nullptr, *lab2 = nullptr;
    int rc;
    if ((rc = alloc(&lab, n))) return rc;
    if (kind && (rc = alloc(&lab2, n))) return rc;
    uint32_t* recs[kMaxChannels] = {nullptr, nullptr, nullptr, nullptr};
    const int W = kind ? kPairWords : kRegionWords;
    for (int c = 0; c < a.K; ++c) if ((rc = alloc(&recs[c], (size_t)n * W))) return rc;
    if (n) {
      DeviceBuffers buf;
      uint32_t *k0, *k1, *i0, *i1, *d_cnt;
      if ((rc = buf.get(&k0, n, false, s)) || (rc = buf.get(&k1, n, false, s)) || (rc = buf.get(&i0, n, false, s)) || (rc = buf.get(&i1, n, false, s)) ||
          (rc = buf.get(&d_cnt, (size_t)world + 1, true, s))) return rc;
      hipLaunchKernelGGL(dest_keys, dim3((n + 255) / 256), dim3(256), 0, s, kind ? a.d_pa : a.d_rlabel, kind ? d_pcut : d_rcut, n, (uint32_t)world, k0, i0, d_cnt);
      size_t tmp = 0;
      GLIA_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, i0, i1, (size_t)n, 0, 16, s));      // stable: the records of a destination keep their (sorted-by-key) order
      char* d_tmp;
      if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, s))) return rc;
      GLIA_HIP_TRY(rocprim::radix_sort_pairs((void*)d_tmp, tmp, k0, k1, i0, i1, (size_t)n, 0, 16, s));
      hipLaunchKernelGGL(permute_rows, dim3((n + 255) / 256), dim3(256), 0, s, kind ? a.d_pa : a.d_rlabel, i1, n, 1, lab);
      if (kind) hipLaunchKernelGGL(permute_rows, dim3((n + 255) / 256), dim3(256), 0, s, a.d_pb, i1, n, 1, lab2);
      for (int c = 0; c < a.K; ++c) {
        const unsigned long long t = (unsigned long long)n * W;
        hipLaunchKernelGGL(permute_rows, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, kind ? a.c_prec[c] : a.c_rrec[c], i1, n, W, recs[c]);
      }
      std::vector<uint32_t> h((size_t)world + 1);
      GLIA_HIP_TRY(hipMemcpyAsync(h.data(), d_cnt, sizeof(uint32_t) * h.size(), hipMemcpyDeviceToHost, s));
      GLIA_HIP_TRY(hipStreamSynchronize(s));
      for (size_t d = 0; d < h.size(); ++d) cnt[d] = h[d];
    }
    if (kind) { out->arr.d_pa = lab; out->arr.d_pb = lab2; for (int c = 0; c < a.K; ++c) out->arr.c_prec[c] = recs[c]; out->arr.d_prec = recs[0]; }
    else { out->arr.d_rlabel = lab; for (int c = 0; c < a.K; ++c) out->arr.c_rrec[c] = recs[c]; out->arr.d_rrec = recs[0]; }
    if (kind) {
      out->vcount.assign((size_t)world + 1, 0);
      if (a.d_pv_off && n) {
        // the value runs follow their pairs into destination order; values per destination = differences of the new offsets
        DeviceBuffers buf2;
        uint32_t *k0, *k1, *i0, *i1, *d_cnt;
        if ((rc = buf2.get(&k0, n, false, s)) || (rc = buf2.get(&k1, n, false, s)) || (rc = buf2.get(&i0, n, false, s)) || (rc = buf2.get(&i1, n, false, s)) ||
            (rc = buf2.get(&d_cnt, (size_t)world + 1, true, s))) return rc;
        hipLaunchKernelGGL(dest_keys, dim3((n + 255) / 256), dim3(256), 0, s, a.d_pa, d_pcut, n, (uint32_t)world, k0, i0, d_cnt);
        size_t tmp = 0;
        GLIA_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, i0, i1, (size_t)n, 0, 16, s));
        char* d_tmp;
        if ((rc = buf2.get(&d_tmp, tmp ? tmp : 16, false, s))) return rc;
        GLIA_HIP_TRY(rocprim::radix_sort_pairs((void*)d_tmp, tmp, k0, k1, i0, i1, (size_t)n, 0, 16, s));
        if ((rc = gather_value_runs(a.d_pv_off, a.d_pv, i1, n, &out->arr.d_pv_off, &out->arr.d_pv, &out->arr.nV, s))) return rc;
        out->owned.push_back(out->arr.d_pv_off); out->owned.push_back(out->arr.d_pv);
        std::vector<unsigned long long> off((size_t)n + 1);
        GLIA_HIP_TRY(hipMemcpy(off.data(), out->arr.d_pv_off, sizeof(unsigned long long) * off.size(), hipMemcpyDeviceToHost));
        uint64_t p0 = 0;
        for (int d = 0; d <= world; ++d) { const uint64_t p1 = p0 + out->pcount[d]; out->vcount[d] = off[p1] - off[p0]; p0 = p1; }
      }
    }
  }
  GLIA_HIP_TRY(hipGetLastError());
  return GLIA_HMT_OK;
}

// a block of records on the device: [R labels | K x R region records | P a | P b | K x P pair records]
struct Block {
  uint32_t* base = nullptr; uint64_t R = 0, P = 0, V = 0; int K = 1;
  unsigned long long* pv_off = nullptr;         // offsets of the value runs, recomputed from the pair counts at the receiver
  static size_t words(uint64_t R, uint64_t P, int K, uint64_t V = 0) { return (size_t)R * (1 + (size_t)K * kRegionWords) + (size_t)P * (2 + (size_t)K * kPairWords) + (size_t)V; }
  float* pv() const { return reinterpret_cast<float*>(prec(K - 1) + P * kPairWords); }
  uint32_t* rlabel() const { return base; }
  uint32_t* rrec(int c) const { return base + R + (size_t)c * R * kRegionWords; }
  uint32_t* pa() const { return base + R * (1 + (size_t)K * kRegionWords); }
  uint32_t* pb() const { return pa() + P; }
  uint32_t* prec(int c) const { return pb() + P + (size_t)c * P * kPairWords; }
  RagArrays view(const RagArrays& like) const {
    RagArrays a; a.R = (int64_t)R; a.P = (int64_t)P; a.K = K;
    a.d_rlabel = rlabel(); a.d_pa = pa(); a.d_pb = pb();
    for (int c = 0; c < K; ++c) { a.c_rrec[c] = rrec(c); a.c_prec[c] = prec(c); a.c_bins[c] = like.c_bins[c]; }
    a.d_rrec = a.c_rrec[0]; a.d_prec = a.c_prec[0];
    if (pv_off) { a.d_pv_off = pv_off; a.d_pv = pv(); a.nV = V; }
    return a;
  }
};
// offsets of a received block's value runs = scan of its pairs' voxel counts
__global__ void block_counts(const uint32_t* prec, uint32_t P, unsigned long long* cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > P) return;
  cnt[i] = i < P ? (unsigned long long)prec[(size_t)i * kPairWords + P_CNT] : 0ull;
}
int block_value_offsets(Block* b, hipStream_t s, std::vector<void*>* scratch) {
  const uint32_t P = (uint32_t)b->P;
  DeviceBuffers buf;
  int rc;
  unsigned long long* cnt;
  if ((rc = buf.get(&cnt, (size_t)P + 1, false, s))) return rc;
  GLIA_HIP_TRY(hipMalloc(&b->pv_off, sizeof(unsigned long long) * ((size_t)P + 1)));
  scratch->push_back(b->pv_off);
  hipLaunchKernelGGL(block_counts, dim3((P + 256) / 256), dim3(256), 0, s, b->prec(0), P, cnt);
  size_t tmp = 0;
  GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, cnt, b->pv_off, 0ull, (size_t)P + 1, rocprim::plus<unsigned long long>(), s));
  char* d_tmp;
  if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, s))) return rc;
  GLIA_HIP_TRY(rocprim::exclusive_scan((void*)d_tmp, tmp, cnt, b->pv_off, 0ull, (size_t)P + 1, rocprim::plus<unsigned long long>(), s));
  GLIA_HIP_TRY(hipStreamSynchronize(s));
  return GLIA_HMT_OK;
}
// transfers that move rows [r0, r0 + nR) / [p0, p0 + nP) of `from` (any layout with per-array contiguity) into block `to`
void add_record_transfers(std::vector<Transfer>* ts, int src, int dst, const RagArrays* from, uint64_t r0, uint64_t nR, uint64_t p0, uint64_t nP, const Block* to, int K,
                          uint64_t v0 = 0, uint64_t nV = 0) {
  auto add = [&](const uint32_t* f, uint64_t first, int words, uint64_t n, uint32_t* t) {
    ts->push_back(Transfer{src, dst, from ? (const void*)(f + first * words) : nullptr, to ? (void*)t : nullptr, (size_t)n * words * sizeof(uint32_t)});
  };
  add(from ? from->d_rlabel : nullptr, r0, 1, nR, to ? to->rlabel() : nullptr);
  for (int c = 0; c < K; ++c) add(from ? from->c_rrec[c] : nullptr, r0, kRegionWords, nR, to ? to->rrec(c) : nullptr);
  add(from ? from->d_pa : nullptr, p0, 1, nP, to ? to->pa() : nullptr);
  add(from ? from->d_pb : nullptr, p0, 1, nP, to ? to->pb() : nullptr);
  for (int c = 0; c < K; ++c) add(from ? from->c_prec[c] : nullptr, p0, kPairWords, nP, to ? to->prec(c) : nullptr);
  if (nV) add(from ? reinterpret_cast<const uint32_t*>(from->d_pv) : nullptr, v0, 1, nV, to ? reinterpret_cast<uint32_t*>(to->pv()) : nullptr);
}

void free_arrays(RagArrays* a) {
  (void)hipFree(a->d_rlabel); (void)hipFree(a->d_pa); (void)hipFree(a->d_pb);
  if (a->d_pv_off) (void)hipFree(a->d_pv_off);
  if (a->d_pv) (void)hipFree(a->d_pv);
  for (int c = 0; c < a->K; ++c) { (void)hipFree(a->c_rrec[c]); (void)hipFree(a->c_prec[c]); }
  *a = RagArrays();
}

}  // namespace

extern "C" {

int glia_hmt_comm_unique_id(void* id128) {
  if (!id128) { set_error("comm_unique_id: NULL"); return GLIA_HMT_ERR_ARG; }
  int rc = g_rccl.load();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
  ncclUniqueId id;
  GLIA_NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return GLIA_HMT_OK;
}

int glia_hmt_comm_create_rccl(glia_hmt_ctx* c, int world, int rank, const void* id128, glia_hmt_comm** out) {
  if (!c || !out || world < 1 || rank < 0 || rank >= world || !id128) { set_error("comm_create_rccl: invalid argument"); return GLIA_HMT_ERR_ARG; }
  int rc = g_rccl.load();
  if (rc) return rc;
  GLIA_HIP_TRY(hipSetDevice(c->device));
  glia_hmt_comm* cm = new glia_hmt_comm;
  cm->ctx = c; cm->world = world; cm->local.assign(1, rank);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&cm->nccl, world, id, rank);
  if (r != ncclSuccess) { set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); delete cm; return GLIA_HMT_ERR_HIP; }
  *out = cm;
  return GLIA_HMT_OK;
}

int glia_hmt_comm_create_local(glia_hmt_ctx* c, int world, glia_hmt_comm** out) {
  if (!c || !out || world < 1 || world > 255) { set_error("comm_create_local: invalid argument"); return GLIA_HMT_ERR_ARG; }
  glia_hmt_comm* cm = new glia_hmt_comm;
  cm->ctx = c; cm->world = world;
  for (int r = 0; r < world; ++r) cm->local.push_back(r);
  *out = cm;
  return GLIA_HMT_OK;
}

void glia_hmt_comm_destroy(glia_hmt_comm* cm) {
  if (!cm) return;
  if (cm->nccl) (void)g_rccl.CommDestroy(cm->nccl);
  delete cm;
}
int glia_hmt_comm_world(const glia_hmt_comm* cm) { return cm ? cm->world : -1; }
int glia_hmt_comm_local_ranks(const glia_hmt_comm* cm, int* ranks, int capacity) {
  if (!cm) return -1;
  for (int i = 0; i < (int)cm->local.size() && ranks && i < capacity; ++i) ranks[i] = cm->local[i];
  return (int)cm->local.size();
}

int glia_hmt_slab_range(int64_t nz, int world, int rank, int64_t* first_plane, int64_t* n_planes, int64_t* z_begin, int64_t* z_end) {
  if (nz < 1 || world < 1 || rank < 0 || rank >= world || world > nz) { set_error("slab_range: invalid argument"); return GLIA_HMT_ERR_ARG; }
  const int64_t base = nz / world, rem = nz % world;
  const int64_t z0 = rank * base + (rank < rem ? rank : rem), z1 = z0 + base + (rank < rem ? 1 : 0);
  const int64_t lo = z0 > 0 ? z0 - 1 : 0, hi = z1 < nz ? z1 + 1 : nz;
  if (first_plane) *first_plane = lo;
  if (n_planes) *n_planes = hi - lo;
  if (z_begin) *z_begin = z0 - lo;
  if (z_end) *z_end = z1 - lo;
  return GLIA_HMT_OK;
}

int glia_hmt_rag_build_distributed(glia_hmt_ctx* c, glia_hmt_comm* cm, const glia_hmt_slab* slabs, int64_t nz_global, int only_contour, int with_values,
                                   int loop_owner, glia_hmt_rag** out, glia_hmt_dist_stats* stats) {
  if (!c || !cm || cm->ctx != c || !slabs || !out || loop_owner < 0 || loop_owner >= cm->world) { set_error("rag_build_distributed: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int world = cm->world, nl = (int)cm->local.size();
  *out = nullptr;
  glia_hmt_dist_stats st;
  memset(&st, 0, sizeof(st));
  int rc = GLIA_HMT_OK;
  // ---- 1. partial maps, cut flags, records in destination order ----
  std::vector<glia_hmt_rag*> part((size_t)nl, nullptr);
  std::vector<Sorted> sorted((size_t)nl);
  std::vector<void*> scratch;
  std::vector<RagArrays> reduced((size_t)nl);
  std::vector<std::vector<Block>> recvA((size_t)nl), recvB;
  auto cleanup = [&]() {
    for (glia_hmt_rag* p : part) if (p) glia_hmt_rag_free(p);
    for (Sorted& x : sorted) x.release();
    for (void* p : scratch) (void)hipFree(p);
    for (RagArrays& a : reduced) if (a.d_rlabel) free_arrays(&a);
  };
  for (int i = 0; i < nl && !rc; ++i) {
    const glia_hmt_slab& sl = slabs[i];
    rc = glia_hmt_rag_build_slab(c, sl.dims_local, sl.z_global_of_plane0, nz_global, sl.z_begin, sl.z_end, sl.d_labels, only_contour, sl.d_pb, sl.cfg, &part[i]);
    if (rc) break;
    const RagArrays& a = part[i]->arr;
    uint8_t *rcut = nullptr, *pcut = nullptr;
    if (hipMalloc(&rcut, (size_t)(a.R ? a.R : 1)) != hipSuccess || hipMalloc(&pcut, (size_t)(a.P ? a.P : 1)) != hipSuccess) { set_error("rag_build_distributed: out of memory"); rc = GLIA_HMT_ERR_HIP; break; }
    scratch.push_back(rcut); scratch.push_back(pcut);
    rc = rag_cut_flags(a, sl.d_labels, sl.dims_local[0], sl.dims_local[1], sl.dims_local[2], sl.z_begin, sl.z_end, rcut, pcut, s);
    // median linkage downstream: the image value of every boundary voxel goes along with its directed pair
    if (!rc && with_values) rc = collect_pair_values(&part[i]->arr, part[i]->slab, s);
    if (!rc) rc = sort_by_destination(part[i]->arr, rcut, pcut, world, s, &sorted[i]);
    st.records += (uint64_t)(a.R + a.P);
  }
  if (rc) { cleanup(); return rc; }
  const int K = part[0]->arr.K;
  // counts of every (source, destination): [world][2 * (world + 1)]
  std::vector<std::vector<uint64_t>> mine((size_t)nl);
  const int ncol = 3 * (world + 1);
  for (int i = 0; i < nl; ++i) {
    mine[i] = sorted[i].rcount;
    mine[i].insert(mine[i].end(), sorted[i].pcount.begin(), sorted[i].pcount.end());
    mine[i].insert(mine[i].end(), sorted[i].vcount.begin(), sorted[i].vcount.end());
  }
  std::vector<uint64_t> cnt;
  if ((rc = all_gather_u64(cm, mine, ncol, &cnt))) { cleanup(); return rc; }
  auto cR = [&](int src, int d) { return cnt[(size_t)src * ncol + d]; };
  auto cP = [&](int src, int d) { return cnt[(size_t)src * ncol + (world + 1) + d]; };
  auto cV = [&](int src, int d) { return cnt[(size_t)src * ncol + 2 * (world + 1) + d]; };
  for (int i = 0; i < nl; ++i) for (int d = 0; d < world; ++d) st.cut_records += cR(cm->local[i], d) + cP(cm->local[i], d);
  // ---- 2. keyed owner exchange of the cut records, reduction at the owner ----
  {
    std::vector<Transfer> ts;
    for (int i = 0; i < nl; ++i) recvA[i].assign((size_t)world, Block());
    for (int src = 0; src < world; ++src) {
      const int li = [&] { for (int i = 0; i < nl; ++i) if (cm->local[i] == src) return i; return -1; }();
      uint64_t r0 = 0, p0 = 0, v0 = 0;
      for (int d = 0; d < world; ++d) {
        const uint64_t nR = cR(src, d), nP = cP(src, d), nV = cV(src, d);
        const int ld = [&] { for (int i = 0; i < nl; ++i) if (cm->local[i] == d) return i; return -1; }();
        Block* to = nullptr;
        if (ld >= 0) {
          Block& b = recvA[ld][src];
          b.R = nR; b.P = nP; b.V = nV; b.K = K;
          if (hipMalloc(&b.base, sizeof(uint32_t) * (Block::words(nR, nP, K, nV) + 1)) != hipSuccess) { set_error("rag_build_distributed: out of memory"); cleanup(); return GLIA_HMT_ERR_HIP; }
          scratch.push_back(b.base);
          to = &b;
        }
        add_record_transfers(&ts, src, d, li >= 0 ? &sorted[li].arr : nullptr, r0, nR, p0, nP, to, K, v0, nV);
        r0 += nR; p0 += nP; v0 += nV;
      }
    }
    uint64_t sent = 0;
    if ((rc = run_transfers(cm, ts, &sent))) { cleanup(); return rc; }
    st.bytes_cut_exchange = sent;
    if (with_values)
      for (int i = 0; i < nl; ++i) for (int src = 0; src < world; ++src) if (recvA[i][src].P && (rc = block_value_offsets(&recvA[i][src], s, &scratch))) { cleanup(); return rc; }
  }
  for (int i = 0; i < nl; ++i) {
    std::vector<RagArrays> parts;
    for (int src = 0; src < world; ++src) if (recvA[i][src].R || recvA[i][src].P) parts.push_back(recvA[i][src].view(part[0]->arr));
    if (parts.empty()) { reduced[i] = RagArrays(); reduced[i].K = K; continue; }
    if ((rc = merge_rag_arrays(parts.data(), (int)parts.size(), &reduced[i], s))) { cleanup(); return rc; }
  }
  // ---- 3. everything once to the loop owner, final reduction by key ----
  std::vector<std::vector<uint64_t>> mine2((size_t)nl);
  for (int i = 0; i < nl; ++i) mine2[i] = {(uint64_t)reduced[i].R, (uint64_t)reduced[i].P, (uint64_t)reduced[i].nV};
  std::vector<uint64_t> cnt2;
  if ((rc = all_gather_u64(cm, mine2, 3, &cnt2))) { cleanup(); return rc; }
  const bool own = cm->is_local(loop_owner);
  std::vector<Block> fin;
  {
    std::vector<Transfer> ts;
    if (own) fin.assign((size_t)2 * world, Block());
    for (int src = 0; src < world; ++src) {
      const int li = [&] { for (int i = 0; i < nl; ++i) if (cm->local[i] == src) return i; return -1; }();
      // interior records: the last destination class of the sorted arrays
      uint64_t r0 = 0, p0 = 0, v0 = 0;
      for (int d = 0; d < world; ++d) { r0 += cR(src, d); p0 += cP(src, d); v0 += cV(src, d); }
      for (int piece = 0; piece < 2; ++piece) {
        const uint64_t nR = piece ? cnt2[(size_t)src * 3] : cR(src, world), nP = piece ? cnt2[(size_t)src * 3 + 1] : cP(src, world);
        const uint64_t nV = piece ? cnt2[(size_t)src * 3 + 2] : cV(src, world);
        Block* to = nullptr;
        if (own) {
          Block& b = fin[(size_t)2 * src + piece];
          b.R = nR; b.P = nP; b.V = nV; b.K = K;
          if (hipMalloc(&b.base, sizeof(uint32_t) * (Block::words(nR, nP, K, nV) + 1)) != hipSuccess) { set_error("rag_build_distributed: out of memory"); cleanup(); return GLIA_HMT_ERR_HIP; }
          scratch.push_back(b.base);
          to = &b;
        }
        const RagArrays* from = li < 0 ? nullptr : (piece ? &reduced[li] : &sorted[li].arr);
        add_record_transfers(&ts, src, loop_owner, from, piece ? 0 : r0, nR, piece ? 0 : p0, nP, to, K, piece ? 0 : v0, nV);
      }
    }
    uint64_t sent = 0;
    if ((rc = run_transfers(cm, ts, &sent))) { cleanup(); return rc; }
    st.bytes_to_loop_owner = sent;
    if (own && with_values)
      for (Block& b : fin) if (b.P && (rc = block_value_offsets(&b, s, &scratch))) { cleanup(); return rc; }
  }
  if (own) {
    std::vector<RagArrays> parts;
    for (const Block& b : fin) if (b.R || b.P) parts.push_back(b.view(part[0]->arr));
    glia_hmt_rag* rag = new glia_hmt_rag(*part[0]);
    rag->arr = RagArrays(); rag->arr.K = K;
    rag->d_folded = nullptr; rag->vol = VolumeRef(); rag->slab = VolumeRef();
    rag->dims[2] = nz_global;
    rag->pass_ms = 0; rag->alg_bytes = 0;
    for (glia_hmt_rag* p : part) { rag->pass_ms += p->pass_ms; rag->alg_bytes += p->alg_bytes; }
    if (!parts.empty()) rc = merge_rag_arrays(parts.data(), (int)parts.size(), &rag->arr, s);
    if (rc) { delete rag; cleanup(); return rc; }
    for (int ch = 0; ch < kMaxChannels; ++ch) rag->arr.c_bins[ch] = part[0]->arr.c_bins[ch];
    *out = rag;
  }
  GLIA_HIP_TRY(hipStreamSynchronize(s));
  cleanup();
  if (stats) *stats = st;
  return GLIA_HMT_OK;
}

}  // extern "C"
