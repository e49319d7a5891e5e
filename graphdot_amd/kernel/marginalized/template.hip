// JIT translation unit of the MI355X marginalized-graph-kernel solver.
// Rendered by _backend_hip.py: ${...} placeholders carry the generated
// microkernel functors, the node/edge struct declarations and one
// extern "C" entry point per requested solver variant.  (Role of the
// reference's graphdot/kernel/marginalized/template.cu.)
#define GD_REAL ${real}
#define GD_WEIGHTED ${weighted}
#include <hip/hip_runtime.h>
#include <numpy_type.h>
#include <fmath.h>
#include <array.h>
#include <frozen_array.h>
#include <basekernel.h>
#include <graph.h>
#include <wave.h>
#include <mgk_solver.h>
#include <mgk_oc.h>
#include <mgk_stream.h>
#include <mgk_mfma.h>

using namespace graphdot::numpy_type;
using namespace graphdot::basekernel;
using graphdot::real_t;

${node_t}
${edge_t}

${node_kernel}
${edge_kernel}
${p_start}

using graph_t = graphdot::graph_t<node_t, edge_t>;
using params_t = graphdot::mgk::params_t<real_t, graph_t, node_kernel_t,
                                         edge_kernel_t, p_start_t>;
static_assert(sizeof(graphdot::graph_header_t) == 64, "graph header layout");
static_assert(sizeof(node_t) == ${node_size}, "node_t layout differs from the host packer");
static_assert(sizeof(edge_t) == ${edge_size}, "edge_t layout differs from the host packer");
static_assert(sizeof(params_t) == ${params_size}, "params_t layout differs from the host packer");
using params_fd_t = graphdot::mgk::params_fd_t<real_t, graph_t, node_kernel_t,
                                               edge_kernel_t, p_start_t>;
static_assert(sizeof(params_fd_t) == ${params_fd_size}, "params_fd_t layout differs from the host packer");

${entry_points}
