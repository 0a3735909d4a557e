// skew.cpp -- Skew Cartesian partitioner (placeholder until implemented).
#include "partition.hpp"
namespace hymls {
int skew_num_subdomains(const Params&) { HYMLS_CHECK(false, -99, "Skew Cartesian partitioner not implemented yet"); return 0; }
void skew_get_groups(const Params&, int, ivec&, std::vector<Group>&) { HYMLS_CHECK(false, -99, "Skew Cartesian partitioner not implemented yet"); }
}
