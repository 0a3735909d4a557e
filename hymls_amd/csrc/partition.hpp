// partition.hpp -- Cartesian partitioner + hierarchical map (host, integer work only).
// Re-designed from the behaviour of the reference's
//   src/HYMLS_CartesianPartitioner.cpp:80-121,224-408   (subdomain grid, GetGroups)
//   src/HYMLS_OverlappingPartitioner.cpp:121-147         (DetectSeparators)
//   src/HYMLS_HierarchicalMap.cpp:120-285                (LinkSeparators, FillComplete)
#pragma once
#include "common.hpp"

namespace hymls {

struct Group {
  int32_t type = -1;
  ivec nodes;  // sorted gids; nodes[0] is the group's V-sum node
};

struct Subdomain {
  ivec interior;               // sorted gids
  std::vector<Group> groups;   // every separator group touching this subdomain
  ivec owned;                  // indices into groups: groups first listed by this subdomain
  std::vector<ivec> linked;        // all groups, linked by equal type >= 0
  std::vector<ivec> owned_linked;  // owned groups, linked by equal type >= 0
  int32_t num_sep() const {
    int32_t n = 0;
    for (auto& g : groups) n += (int32_t)g.nodes.size();
    return n;
  }
};

struct HierMap {
  std::vector<Subdomain> sd;
  int64_t ngid = 0;
};

// GetGroups for one subdomain of the Cartesian partitioner (no filtering).
void cartesian_get_groups(const Params& p, int sd, ivec& interior, std::vector<Group>& groups);
int cartesian_num_subdomains(const Params& p);

// all subdomains + FillComplete semantics; `present` (size ngid) marks gids that
// exist on this level (nullptr: all).  `cand` (size = number of subdomains, nullptr: all) restricts
// the work to a subset of the subdomains (sharded runs: the rank's own subdomains plus a halo);
// the others stay empty, and "first listed by" is then relative to the subset.
HierMap build_hiermap(const Params& p, const std::vector<char>* present, const std::vector<char>* cand = nullptr);
int num_subdomains(const Params& p);
// reference corner of a subdomain in cell coordinates (may lie outside the grid for the clipped
// diamonds of the Skew Cartesian partitioner); every node of the subdomain and of its separators is
// within [pos - sx - 1, pos + sx + 1] in each direction
void sd_position(const Params& p, int sd, int& x, int& y, int& z);

}  // namespace hymls
