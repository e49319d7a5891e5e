// Device image of one graph, as packed by kernel/marginalized/_devicegraph.py.
//
// The reference stores adjacency as compact 8x8 "octiles" + 64-bit masks
// (graphdot/cpp/graph.h:8-33) because its solver walks tile pairs with
// 32-lane half-tiles and scatters with float atomics.  The MI355X solver is
// atomic-free: it needs, per graph, plain CSR of the *directed* nonzeros with
// the nodes renumbered by descending adjacency count, so that any run of
// consecutive rows (tasks) handled by one wave instruction has near-equal
// trip counts and the wave-uniform maximum is known from its first row.
#ifndef GRAPHDOT_HIP_GRAPH_H_
#define GRAPHDOT_HIP_GRAPH_H_
#include <cstdint>

namespace graphdot {

struct nz_t {            // one directed nonzero of the adjacency matrix
    std::uint16_t i, j;  // row (source), column (target), new numbering
};

template<class Node, class Edge> struct graph_t {  // 56 bytes
    using node_t = Node;
    using edge_t = Edge;
    std::int32_t n_node;
    std::int32_t n_nz;             // directed nonzeros (self loop counted once)
    float const *degree;           // [n_node] sum of incident weights, 0 -> 1
    node_t const *node;            // [n_node] AoS node labels
    std::uint16_t const *rowptr;   // [n_node + 1] CSR row starts
    nz_t const *nz;                // [n_nz] CSR order
    edge_t const *edge;            // [n_nz] AoS edge labels (incl. weight)
    std::uint16_t const *perm;     // [n_node] new id -> caller's node id
};

}  // namespace graphdot
#endif
