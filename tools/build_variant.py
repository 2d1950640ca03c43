#!/usr/bin/env python3
"""Experiment builds of the library beside the product: tools/build_variant.py NAME -DFLAG ... ->
ntpoly_amd/libntpoly_amd_NAME.so (objects in ntpoly_amd/build_NAME/).  Selected at run time with
NTPOLY_AMD_LIB=ntpoly_amd/libntpoly_amd_NAME.so (ntpoly_amd/capi.py); A/B runs of kernel variants in one GPU call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntpoly_amd import _build

name, flags = sys.argv[1], sys.argv[2:]
lib = os.path.join(_build.HERE, "libntpoly_amd_%s.so" % name)
print(_build.build(lib=lib, objdir=os.path.join(_build.HERE, "build_" + name), flags=_build.FLAGS + flags))
