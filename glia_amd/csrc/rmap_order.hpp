// glia_amd/csrc/rmap_order.hpp -- iteration order of the reference's region map (host code).
#pragma once
#include <stdint.h>
#include <vector>

namespace glia {

// rank[i] = position of leaf i (label labels[i], first voxel index first[i]) in the iteration order of the reference's
// TRegionMap: replay of genPointMap (util/struct.hxx:77-92: cmap in first-raster-occurrence order, pmap in cmap iteration
// order) and TRegionMap::init (type/region_map.hxx:79-95: emplace in pmap iteration order).  labels ascending, distinct.
// mode: 0 = automatic (array emulation of the libstdc++ hashtable if a start-up probe finds it identical to the real
// container, else the container), 1 = the container, 2 = the emulation (tests).
void rmap_ranks(const std::vector<uint32_t>& labels, const std::vector<long long>& first, std::vector<uint32_t>* rank, int mode = 0);
// the same with the leaves already ordered by their first voxel (byFirst[j] = leaf index; the device sorts them)
void rmap_ranks_ordered(const std::vector<uint32_t>& labels, const std::vector<uint32_t>& byFirst, std::vector<uint32_t>* rank, int mode = 0);

}  // namespace glia
