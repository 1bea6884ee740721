#!/usr/bin/env python3
"""Emit the device headers as C++ raw string literals (adjacent literals concatenate) for hiprtc."""
import sys

out = []
for path in sys.argv[1:]:
    text = open(path).read().replace("#pragma once", "")
    # MSVC-free toolchain, but keep each literal comfortably small
    for k in range(0, len(text), 8000):
        out.append('R"QHIPSRC(' + text[k:k + 8000] + ')QHIPSRC"')
print("\n".join(out))
