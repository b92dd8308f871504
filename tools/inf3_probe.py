#!/usr/bin/env python3
"""GPU: the base inference configuration of tools/variants_bench.py three times in one process (is a later model instance slower than the
first?  profiles/README.md, round 6: sporadically 23-27 ms instead of 18.2 with identical kernel times -- host-side, shared pod)."""
import sys
sys.argv=["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("vb", "tools/variants_bench.py"); vb = importlib.util.module_from_spec(spec); spec.loader.exec_module(vb)
for i in range(3):
    vb.run("base", "v1", False)
