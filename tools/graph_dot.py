#!/usr/bin/env python
"""Summarise a graph dump of this HIP runtime (DEBUG_HIP_GRAPH_DOT_PRINT=1 writes graph_<pid>_dot_print_<n> into the working directory when a captured
graph is instantiated): one line per node in creation order -- kernel, the internal stream the runtime assigned it to, whether its completion is
signalled to another stream, its parents.  The runtime enqueues the nodes in this order; a node whose parent sits on another stream waits for
everything that stream was handed before it (profiles/r03_graph_capture_order.txt), so the order of capture decides what overlaps.
usage: python tools/graph_dot.py <dump>"""
import re
import sys

s = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"graph_(\d+)_node_(\d+)"\[style="\w+"shape="\w+"label="(\d+)\n([^\n]*)\n(?:\(([^\n]*)\)\n)?StreamId:(\d+)\nSignalIsRequired: (\w+)', s):
    name = m.group(4)
    mm = re.search(r"nsig\d+(k_\w+?)E", name) or re.search(r"_Z\d+(k_[a-z_0-9]+?)\d", name)
    nodes[(int(m.group(1)), int(m.group(2)))] = ((mm.group(1) if mm else name)[:30], int(m.group(6)), m.group(7) == "true")
par = {}
for g, a, g2, b in re.findall(r'"graph_(\d+)_node_(\d+)" -> "graph_(\d+)_node_(\d+)"', s):
    par.setdefault((int(g2), int(b)), []).append(int(a))
for k in sorted(nodes):
    name, stream, sig = nodes[k]
    print(f"{k[1]:3d}  stream {stream}  {'signals' if sig else '       '}  {name:30s}  parents {par.get(k, [])}")
