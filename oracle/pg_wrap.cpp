// TEST INFRASTRUCTURE ONLY.  C entry points around the REFERENCE's own host-side PointGroup ops -- the rule-book builder behind
// pointgroup_ops.voxelization_idx (PointGroup/lib/pointgroup_ops/src/voxelize/voxelize.cpp:34-152) and the queue BFS behind
// pointgroup_ops.bfs_cluster (src/bfs_cluster/bfs_cluster.cpp:33-91).  The extension as a whole cannot be built here (its other
// translation units are CUDA), but these functions work on raw pointers and need only datatype.h / datatype.cpp, which are compiled
// where they lie.  oracle/build_ref.py copies exactly those line ranges from /root/reference AT BUILD TIME into the git-ignored
// oracle/_ref/ and compiles this file around them; google-sparsehash is replaced by the 10-line stand-in oracle/pg_shim.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <limits>
#include <datatype/datatype.h>
#include <datatype/datatype.cpp>
#include "_ref/pg_voxelize_extract.inc"
#include "_ref/pg_bfs_extract.inc"

// voxelize_idx (voxelize.cpp:10-31) with the at::Tensor resize / zero calls spelled out on caller buffers.
// coords (n, ncol) int64, ncol 3 or 4.  Returns nActive; *max_active_out = maxActive.  Two-call protocol: output_map / output_coords
// may be NULL to query the sizes first.
extern "C" int ref_voxelize_idx(const long* coords, int n, int ncol, int batch_size, int mode, int* input_map, int* max_active_out,
                                int* output_map, long* output_coords) {
  RuleBook rules;
  SparseGrids<3> sgs;
  Int n_active = 0;
  std::vector<long> c(coords, coords + (size_t)n * ncol);
  Int max_active = voxelize_inputmap<3>(sgs, input_map, rules, n_active, c.data(), n, ncol, batch_size, mode);
  *max_active_out = max_active;
  if (output_map && output_coords && n_active > 0) {
    std::memset(output_map, 0, sizeof(int) * (size_t)n_active * (max_active + 1));
    std::memset(output_coords, 0, sizeof(long) * (size_t)n_active * ncol);
    if (ncol == 4) voxelize_outputmap<3>(c.data(), output_coords, output_map, &rules[1][0], n_active, max_active);
    else {   // voxelize_outputmap strides by dimension + 1: the reference only calls it with batch-indexed coordinates (predicter.py:285)
      return -1;
    }
  }
  return n_active;
}

// bfs_cluster (bfs_cluster.cpp:98-116).  Returns nCluster; *sum_out = sumNPoint.  cluster_idxs (sumNPoint,2) / cluster_offsets
// (nCluster+1) may be NULL to query the sizes first.
extern "C" int ref_bfs_cluster(int* semantic_label, int* ball_query_idxs, int* start_len, int n, int threshold, int* sum_out,
                               int* cluster_idxs, int* cluster_offsets) {
  ConnectedComponents ccs;
  const int sum = get_clusters(semantic_label, ball_query_idxs, start_len, n, threshold, ccs);
  *sum_out = sum;
  if (cluster_idxs && cluster_offsets) {
    std::memset(cluster_idxs, 0, sizeof(int) * (size_t)sum * 2);
    std::memset(cluster_offsets, 0, sizeof(int) * (ccs.size() + 1));
    fill_cluster_idxs_(ccs, cluster_idxs, cluster_offsets);
  }
  return (int)ccs.size();
}
