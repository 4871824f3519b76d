"""Diagnostic: a measurement build lib/libfx_hip_<name>.so of the test-hooks library with extra -D flags.  usage: tools/build_variant.py NAME [-DFLAG ...]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_extraction_amd import build
name=sys.argv[1]; defs=sys.argv[2:]
print(build.build_variant(name, ["-DFX_TEST_HOOKS"]+defs, force=True))
