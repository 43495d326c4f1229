// Fill-reducing orderings for the sparse engine (host only, no HIP): what cholmod.symbolic's ordering step
// (cholmod_analyze_p, reference src/C/cholmod.c:309; AMD / nested dissection inside SuiteSparse, not vendored) does
// for the reference.  The permutation only changes the amount of fill and the shape of the supernodal tree, never the
// solution of S x = b.
#pragma once
#include <cstdint>
#include <vector>

namespace mi355kkt {

using Graph = std::vector<std::vector<int>>;   // symmetric adjacency lists, sorted, no self loops

struct OrderingInfo {
    int method = 0;          // 1 = nested dissection, 2 = approximate minimum degree
    int64_t nnz_nd = 0, nnz_amd = 0;        // nnz(L) of the two candidates (0 = not computed)
    double flops_nd = 0.0, flops_amd = 0.0; // sum_j (1 + colcount_j)^2
    int levels_nd = 0, levels_amd = 0;      // height of the supernodal elimination tree
    std::vector<int> parent;                // elimination tree of the chosen ordering (postordered: parent[j] > j)
    std::vector<int64_t> colcount;          // its column counts (entries below the diagonal)
};

// order[new] = old, postordered along the elimination tree.  `method`: 0 = choose (both candidates, lower cost),
// 1 = nested dissection only, 2 = approximate minimum degree only.
void fill_reducing_ordering(const Graph& adj, std::vector<int>& order, int method, OrderingInfo* info);

// elimination tree and column counts (entries below the diagonal) of the Cholesky factor of the graph permuted by order
void etree_and_counts(const Graph& adj, const std::vector<int>& order, std::vector<int>& parent, std::vector<int64_t>& cc);

// column counts by the O(nnz(L)) row-subtree walk (reference implementation for the tests of etree_and_counts)
void column_counts_by_row_subtrees(const Graph& adj, const std::vector<int>& order, const std::vector<int>& parent,
                                   std::vector<int64_t>& cc);

// relaxed supernode partition of a postordered elimination tree (first column of every supernode, supernode of every column)
void relaxed_supernodes(const std::vector<int>& parent, const std::vector<int64_t>& cc, std::vector<int>& sn_first,
                        std::vector<int>& sn_of);

// approximate minimum degree on the subgraph induced by `nodes`; `halo` nodes (disjoint from nodes) take part in the
// degrees but are never eliminated.  Writes |nodes| ids to out (elimination order).  `local` is an n-sized scratch
// array filled with -1 on entry and on exit.
void amd_order(const Graph& adj, const std::vector<int>& nodes, const std::vector<int>& halo, std::vector<int>& local,
               int* out);

}  // namespace mi355kkt
