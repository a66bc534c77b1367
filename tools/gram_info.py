"""Tile program of the model the fused Gram runs on (link-merged / regrouped): python tools/gram_info.py [robot] [k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flobaroid_amd._lib import Engine
from flobaroid_amd.topology import Topology
name = sys.argv[1] if len(sys.argv) > 1 else "walkman_apriori"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
topo = Topology.load(os.path.join(ROOT, f"flobaroid_amd/robots/{name}.topology.json"))
eng = Engine(topo, floating=True)
print(eng.link_merge_info(), eng.gram_program_info(k))
