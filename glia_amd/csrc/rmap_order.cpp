// glia_amd/csrc/rmap_order.cpp -- which of two regions is "region 0" of an initial edge when everything else ties is
// decided, in the reference, by the iteration order of std::unordered_map<Label, ...> (TRegionMap, type/region_map.hxx:79-95,
// filled from genPointMap's maps, util/struct.hxx:77-92).  That order is a property of libstdc++'s hashtable: identity hash,
// bucket = key mod a prime, one forward list of all nodes, a node of an empty bucket goes to the FRONT of the list, a node of
// a used bucket goes behind the bucket's first-node predecessor, a rehash re-threads the list in its current order by the
// same two rules.  Replaying 3 x R insertions through the real container costs ~60 ms at R = 262 144 (node allocations,
// pointer chasing); the array emulation below does the same moves on indices, asks the library's own
// std::__detail::_Prime_rehash_policy when and to what size to grow, and is checked against the real container on a probe
// set the first time it is used (any difference -> the real container is used for good).
#include "rmap_order.hpp"

#include <algorithm>
#include <unordered_map>

namespace glia {
namespace {

constexpr uint32_t NIL = 0xFFFFFFFFu, BB = 0xFFFFFFFEu;      // no node / the before-begin sentinel

struct ReplayTable {
  std::vector<uint32_t> key, val, next, bkt, tmp;
  uint32_t head = NIL;
  size_t nb = 1, count = 0;
  std::__detail::_Prime_rehash_policy pol;
  explicit ReplayTable(size_t expect) { key.reserve(expect); val.reserve(expect); next.reserve(expect); bkt.assign(1, NIL); }
  static size_t bucket_of(uint32_t k, size_t n) { return n <= 0xFFFFFFFFull ? (size_t)(k % (uint32_t)n) : (size_t)k % n; }   // (32-bit division is the cheap one)
  uint32_t nx(uint32_t x) const { return x == BB ? head : next[x]; }
  void set_nx(uint32_t x, uint32_t v) { if (x == BB) head = v; else next[x] = v; }
  void rehash(size_t n) {                                     // _Hashtable::_M_rehash_aux(n, unique keys)
    tmp.assign(n, NIL);
    uint32_t p = head;
    head = NIL;
    size_t bbegin = 0;
    while (p != NIL) {
      const uint32_t following = next[p];
      const size_t b = bucket_of(key[p], n);
      if (tmp[b] == NIL) {
        next[p] = head; head = p; tmp[b] = BB;
        if (next[p] != NIL) tmp[bbegin] = p;
        bbegin = b;
      } else { const uint32_t before = tmp[b]; next[p] = nx(before); set_nx(before, p); }
      p = following;
    }
    bkt.swap(tmp);
    nb = n;
  }
  void insert_new(uint32_t k, uint32_t v) {                             // _M_insert_unique_node for a key known to be absent
    const auto grow = pol._M_need_rehash(nb, count, 1);
    if (grow.first) rehash(grow.second);
    const size_t b = bucket_of(k, nb);
    const uint32_t node = (uint32_t)key.size();
    key.push_back(k); val.push_back(v); next.push_back(NIL);
    if (bkt[b] != NIL) { const uint32_t before = bkt[b]; next[node] = nx(before); set_nx(before, node); }
    else {
      next[node] = head; head = node;
      if (next[node] != NIL) bkt[bucket_of(key[next[node]], nb)] = node;
      bkt[b] = BB;
    }
    ++count;
  }
};

void order_by_first(const std::vector<long long>& first, std::vector<uint32_t>* by) {
  const size_t R = first.size();
  by->resize(R);
  for (size_t i = 0; i < R; ++i) (*by)[i] = (uint32_t)i;
  std::sort(by->begin(), by->end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
}

// the iteration order of the region map as a list of leaf indices
void replay_container(const std::vector<uint32_t>& labels, const std::vector<uint32_t>& byFirst, std::vector<uint32_t>* out) {
  std::unordered_map<uint32_t, size_t> cmap;
  for (uint32_t i : byFirst) cmap[labels[i]] = 1;
  std::unordered_map<uint32_t, int> pmap;
  for (auto const& cp : cmap) pmap[cp.first] = 0;
  std::unordered_map<uint32_t, int> rmap;
  for (auto const& pp : pmap) rmap.emplace(pp.first, 0);
  out->clear(); out->reserve(labels.size());
  for (auto const& rp : rmap)                                  // labels are ascending: leaf index by binary search
    out->push_back((uint32_t)(std::lower_bound(labels.begin(), labels.end(), rp.first) - labels.begin()));
}

void replay_emulated(const std::vector<uint32_t>& labels, const std::vector<uint32_t>& byFirst, std::vector<uint32_t>* out) {
  const size_t R = labels.size();
  ReplayTable cmap(R), pmap(R), rmap(R);
  for (uint32_t i : byFirst) cmap.insert_new(labels[i], i);
  for (uint32_t p = cmap.head; p != NIL; p = cmap.next[p]) pmap.insert_new(cmap.key[p], cmap.val[p]);
  for (uint32_t p = pmap.head; p != NIL; p = pmap.next[p]) rmap.insert_new(pmap.key[p], pmap.val[p]);
  out->clear(); out->reserve(R);
  for (uint32_t p = rmap.head; p != NIL; p = rmap.next[p]) out->push_back(rmap.val[p]);
}

bool emulation_matches_container() {
  std::vector<uint32_t> labels;
  std::vector<long long> first;
  uint64_t s = 0x2545F4914F6CDD1Dull;
  uint32_t l = 0;
  for (int i = 0; i < 30000; ++i) {                            // passes a dozen growth steps; sparse and dense label ranges
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    l += 1u + (uint32_t)(s % (i < 15000 ? 3u : 100000u));
    labels.push_back(l); first.push_back((long long)(s >> 20));
  }
  std::vector<uint32_t> by, a, b;
  order_by_first(first, &by);
  replay_container(labels, by, &a);
  replay_emulated(labels, by, &b);
  return a == b;
}

}  // namespace

void rmap_ranks_ordered(const std::vector<uint32_t>& labels, const std::vector<uint32_t>& byFirst, std::vector<uint32_t>* rank, int mode) {
  static const bool emulation_ok = emulation_matches_container();
  std::vector<uint32_t> seq;
  if (mode == 2 || (mode == 0 && emulation_ok)) replay_emulated(labels, byFirst, &seq);
  else replay_container(labels, byFirst, &seq);
  rank->assign(labels.size(), 0);
  for (size_t n = 0; n < seq.size(); ++n) (*rank)[seq[n]] = (uint32_t)n;
}

void rmap_ranks(const std::vector<uint32_t>& labels, const std::vector<long long>& first, std::vector<uint32_t>* rank, int mode) {
  std::vector<uint32_t> by;
  order_by_first(first, &by);
  rmap_ranks_ordered(labels, by, rank, mode);
}

}  // namespace glia
