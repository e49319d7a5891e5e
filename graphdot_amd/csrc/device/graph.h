// Device image of one graph, as packed by kernel/marginalized/_devicegraph.py.
//
// The reference stores adjacency as compact 8x8 "octiles" + 64-bit masks
// (graphdot/cpp/graph.h:8-33) because its solver walks tile pairs with
// 32-lane half-tiles and scatters with float atomics.  The MI355X solver is
// atomic-free: it needs, per graph, plain CSR of the *directed* nonzeros with
// the nodes renumbered by descending adjacency count, so that any run of
// consecutive rows (tasks) handled by one wave instruction has near-equal
// trip counts and the wave-uniform maximum is known from its first row.
//
// The solver copies the packed image of each graph of a pair into LDS (the
// image [degree .. perm] is contiguous) and builds its view on that copy.
// All graphs of a call live in one arena allocation; a header stores byte
// offsets from the arena base (which is a kernel argument, so the compiler
// knows every derived pointer is in global memory and emits global_load /
// s_load rather than flat_load).
#ifndef GRAPHDOT_HIP_GRAPH_H_
#define GRAPHDOT_HIP_GRAPH_H_
#include <cstdint>

namespace graphdot {

struct nz_t {            // one directed nonzero of the adjacency matrix
    std::uint16_t i, j;  // row (source), column (target), new numbering
};

struct graph_header_t {  // 64 bytes, see _devicegraph.HEADER_DTYPE
    std::int32_t n_node;
    std::int32_t n_nz;   // directed nonzeros (self loop counted once)
    std::uint32_t degree, node, rowptr, nz, edge, perm;  // byte offsets
    // nodes per adjacency count 0..14 (hist[15]: 15 and above).  Nodes are
    // stored by descending count, so this is all the owner-computes solver
    // needs to lay out its degree-pair rectangles (mgk_oc.h) -- read with the
    // header into scalar registers instead of counted per pair with ballots.
    std::uint16_t hist[16];
};

template<class Node, class Edge> struct graph_t {
    using node_t = Node;
    using edge_t = Edge;
    int n_node, n_nz;
    float const *degree;           // [n_node] sum of incident weights, 0 -> 1
    node_t const *node;            // [n_node] AoS node labels
    std::uint16_t const *rowptr;   // [n_node + 1] CSR row starts
    nz_t const *nz;                // [n_nz] CSR order
    edge_t const *edge;            // [n_nz] AoS edge labels (incl. weight)
    std::uint16_t const *perm;     // [n_node] new id -> caller's node id

    __device__ __forceinline__ graph_t(char const *arena, graph_header_t const &h)
        : n_node(h.n_node), n_nz(h.n_nz),
          degree(reinterpret_cast<float const *>(arena + h.degree)),
          node(reinterpret_cast<node_t const *>(arena + h.node)),
          rowptr(reinterpret_cast<std::uint16_t const *>(arena + h.rowptr)),
          nz(reinterpret_cast<nz_t const *>(arena + h.nz)),
          edge(reinterpret_cast<edge_t const *>(arena + h.edge)),
          perm(reinterpret_cast<std::uint16_t const *>(arena + h.perm)) {}
};

}  // namespace graphdot
#endif
