"""NetworkX <-> Graph adaptors (reference: ``graphdot/graph/_from_networkx.py``
and ``_to_networkx.py``)."""
import networkx as nx
from ..minipandas import DataFrame


def _common_keys(items, what):
    keys = None
    for ident, attr in items:
        k = sorted(attr.keys())
        if keys is None:
            keys = k
        elif k != keys:
            raise TypeError(f'{what} {ident} attributes {list(attr.keys())} '
                            f'inconsistent with {keys}')
    return keys or []


def _from_networkx(cls, graph, weight=None):
    labels = list(graph.nodes)
    if (not all(isinstance(x, int) for x in labels)
            or min(labels, default=0) < 0
            or max(labels, default=-1) + 1 != len(labels)):
        graph = nx.convert_node_labels_to_integers(graph)

    title = graph.graph.get('title', '')

    node_keys = _common_keys(graph.nodes.items(), 'Node')
    nodes = DataFrame({'!i': range(len(graph.nodes))})
    for key in node_keys:
        nodes[key] = [attr[key] for attr in graph.nodes.values()]

    if len(graph.edges) == 0:
        raise RuntimeError(f'Graph {graph} has no edges.')
    edge_keys = _common_keys(graph.edges.items(), 'Edge')
    edges = DataFrame()
    edges['!i'], edges['!j'] = zip(*graph.edges.keys())
    if weight is not None:
        edges['!w'] = [attr[weight] for attr in graph.edges.values()]
    for key in edge_keys:
        if key != weight:
            edges[key] = [attr[key] for attr in graph.edges.values()]

    return cls(nodes=nodes, edges=edges, title=title)


def _to_networkx(graph):
    g = nx.Graph(title=graph.title)
    node_cols = [c for c in graph.nodes.columns if c != '!i']
    for k, i in enumerate(graph.nodes['!i']):
        g.add_node(int(i), **{c: graph.nodes[c][k] for c in node_cols})
    edge_cols = [c for c in graph.edges.columns if c not in ('!i', '!j')]
    for k, (i, j) in enumerate(zip(graph.edges['!i'], graph.edges['!j'])):
        g.add_edge(int(i), int(j),
                   **{c: graph.edges[c][k] for c in edge_cols})
    return g
