// Device image of one graph, as packed by kernel/marginalized/_devicegraph.py.
//
// The reference stores adjacency as compact 8x8 "octiles" + 64-bit masks
// (graphdot/cpp/graph.h:8-33) because its solver walks tile pairs with
// 32-lane half-tiles.  The MI355X solver keeps the product-graph nonzeros in
// registers and needs, per graph, only the list of *directed* nonzeros
// (source, target, label[, weight]); they are ordered so that any 8 (or 16,
// 32, 64) consecutive entries have distinct sources whenever the graph has
// that many nodes, which makes the LDS scatter of a 64-lane tile of
// (nonzero of G1) x (nonzero of G2) pairs free of same-address collisions.
#ifndef GRAPHDOT_HIP_GRAPH_H_
#define GRAPHDOT_HIP_GRAPH_H_
#include <cstdint>

namespace graphdot {

struct nz_t {            // one directed nonzero of the adjacency matrix
    std::uint16_t i, j;  // row (source), column (target)
};

template<class Node, class Edge> struct graph_t {  // 40 bytes
    using node_t = Node;
    using edge_t = Edge;
    std::int32_t n_node;
    std::int32_t n_nz;      // directed nonzeros (self loop counted once)
    float const *degree;    // [n_node] sum of incident weights, 0 -> 1
    node_t const *node;     // [n_node] AoS node labels, indexed by node id
    nz_t const *nz;         // [n_nz]
    edge_t const *edge;     // [n_nz] AoS edge labels (incl. weight if weighted)
};

}  // namespace graphdot
#endif
