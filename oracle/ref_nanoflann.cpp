// ref_nanoflann.cpp — thin extern "C" shim around the REFERENCE's own kd-tree
// (/root/reference/include/nanoflann.hpp + KDTreeVectorOfVectorsAdaptor.h, compiled where they lie,
// unmodified).  Test infrastructure only: it pins the neighbour-set semantics of
// MultirotorSimulator::handleCollisions (src/multirotor_simulator.cpp:303-328) for the oracle and the
// GPU spatial hash.  Output goes to oracle/_ref/ (git-ignored, travels to the GPU box as a built .so).
//
// The reference instantiates KDTreeVectorOfVectorsAdaptor<std::vector<Eigen::VectorXd>, double>
// (multirotor_simulator.cpp:22,309); Eigen is absent in this image, so the same template is
// instantiated over std::vector<std::vector<double>>, which offers the identical [i][d] interface.
#include <KDTreeVectorOfVectorsAdaptor.h>

#include <cstdint>
#include <vector>

typedef std::vector<std::vector<double>>                        vv_t;
typedef KDTreeVectorOfVectorsAdaptor<vv_t, double>              kd_tree_t;

extern "C" {

// For every point i: RadiusResultSet<double,int>(radius) + findNeighbors, exactly as :321-328.
// offsets has n+1 entries; idx/d2 receive up to cap results (query-major, kd-tree traversal order).
// Returns the total number of results (may exceed cap: call again with a larger buffer).
int64_t ref_nf_radius_all(const double* pts, int32_t n, double radius, int32_t leaf_max_size, int64_t* offsets,
                          int32_t* idx, double* d2, int64_t cap) {
  vv_t poses((size_t)n, std::vector<double>(3));
  for (int32_t i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) poses[(size_t)i][(size_t)d] = pts[3 * (size_t)i + d];
  kd_tree_t mat_index(3, poses, leaf_max_size);
  std::vector<nanoflann::ResultItem<int, double>> indices_dists;
  int64_t total = 0;
  for (int32_t i = 0; i < n; i++) {
    nanoflann::RadiusResultSet<double, int> resultSet(radius, indices_dists);
    mat_index.index->findNeighbors(resultSet, &poses[(size_t)i][0]);
    if (offsets) offsets[i] = total;
    for (size_t j = 0; j < resultSet.m_indices_dists.size(); j++) {
      if (total < cap) {
        if (idx) idx[total] = resultSet.m_indices_dists[j].first;
        if (d2) d2[total] = resultSet.m_indices_dists[j].second;
      }
      total++;
    }
  }
  if (offsets) offsets[n] = total;
  return total;
}

// Timing helper for the cpu_baseline of the collision pass: build + n queries, returns result count.
int64_t ref_nf_build_and_count(const double* pts, int32_t n, double radius, int32_t leaf_max_size) {
  return ref_nf_radius_all(pts, n, radius, leaf_max_size, nullptr, nullptr, nullptr, 0);
}
}
